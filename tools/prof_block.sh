export TUNINGS=${TUNINGS:-sparse}
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/profblk -- python3 $GRAFT_REPO_ROOT/tools/bench_block.py > $GRAFT_REPO_ROOT/gpurun_out/profblk.log 2>&1
tail -3 $GRAFT_REPO_ROOT/gpurun_out/profblk.log
