python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "spmm or transpos" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_t
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_t -- python3 $GRAFT_REPO_ROOT/tools/prof_ops.py transpose 10 > $GRAFT_REPO_ROOT/gpurun_out/prof_t.log 2>&1
grep -h "spt::" $GRAFT_REPO_ROOT/gpurun_out/prof_t/*/*kernel_stats.csv | cut -c1-60,120-240
