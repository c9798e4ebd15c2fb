cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" != "default" ]; then export SPT_HIP_LIBRARY=$GRAFT_REPO_ROOT/spt-proto_amd/lib/libspt_hip_$v.so; else unset SPT_HIP_LIBRARY; fi
  (cd $GRAFT_REPO_ROOT && python -m pytest tests/test_gpu_fused_attention.py -m gpu -x -q 2>&1 | tail -1)
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_fa_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_fa_$v -- python3 $GRAFT_REPO_ROOT/tools/prof_ops.py fused 10 > $GRAFT_REPO_ROOT/gpurun_out/prof_fa_$v.log 2>&1
  grep -h "sparse_attention" $GRAFT_REPO_ROOT/gpurun_out/prof_fa_$v/*/*kernel_stats.csv | awk -F'"' '{print substr($2,11,60), $3}' | cut -c1-110
done
