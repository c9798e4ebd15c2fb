cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export SPT_HIP_LIBRARY=$GRAFT_REPO_ROOT/spt-proto_amd/lib/libspt_hip_$v.so
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_t_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_t_$v -- python3 $GRAFT_REPO_ROOT/tools/prof_ops.py transpose 10 > $GRAFT_REPO_ROOT/gpurun_out/prof_t_$v.log 2>&1
  echo $v; grep -h "spmm_t64" $GRAFT_REPO_ROOT/gpurun_out/prof_t_$v/*/*kernel_stats.csv | cut -c1-40,120-200
done
