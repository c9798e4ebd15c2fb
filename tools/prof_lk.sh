python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "lookup" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_lk
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_lk -- python3 $GRAFT_REPO_ROOT/tools/prof_ops.py lookup 10 > $GRAFT_REPO_ROOT/gpurun_out/prof_lk.log 2>&1
grep -h "lookup" $GRAFT_REPO_ROOT/gpurun_out/prof_lk/*/*kernel_stats.csv | cut -c1-50,100-220
