python tools/bench_long.py 2>/dev/null
export SPARSE_ONLY=1
cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_long && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_long -- python3 $GRAFT_REPO_ROOT/tools/bench_long.py > $GRAFT_REPO_ROOT/gpurun_out/prof_long.log 2>&1
