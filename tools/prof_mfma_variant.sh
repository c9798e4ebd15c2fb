# rocprofv3 kernel stats of tools/time_mfma.py under a variant library: bash tools/prof_mfma_variant.sh NAME
export SPT_HIP_LIBRARY=$GRAFT_REPO_ROOT/spt-proto_amd/lib/variants/libspt_hip_$1.so
cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_mfma_$1 && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_mfma_$1 -- python3 $GRAFT_REPO_ROOT/tools/time_mfma.py > $GRAFT_REPO_ROOT/gpurun_out/prof_mfma_$1.log 2>&1
