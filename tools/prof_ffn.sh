cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/profffn && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/profffn -- python3 $GRAFT_REPO_ROOT/tools/prof_ffn.py > $GRAFT_REPO_ROOT/gpurun_out/profffn.log 2>&1
echo done
