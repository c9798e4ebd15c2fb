python -m pytest tests/test_gpu_ffn.py -m gpu -x -q -k "route" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/profffn && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/profffn -- python3 $GRAFT_REPO_ROOT/tools/prof_ffn.py > $GRAFT_REPO_ROOT/gpurun_out/profffn.log 2>&1
grep -h "route_topk\|rows_combine" $GRAFT_REPO_ROOT/gpurun_out/profffn/*/*kernel_stats.csv | cut -c1-60,100-160
