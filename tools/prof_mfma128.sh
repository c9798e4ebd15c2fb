# rocprofv3 kernel stats of the matrix-core attention kernels at d_head 128 (N=8, S=512, H=32)
cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_mfma128 && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_mfma128 -- python3 $GRAFT_REPO_ROOT/tools/time_mfma.py 8 512 32 128 > $GRAFT_REPO_ROOT/gpurun_out/prof_mfma128.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_mfma128.log
