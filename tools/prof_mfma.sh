# rocprofv3 kernel trace of tools/time_mfma.py (prepare / forward / backward of the MFMA path)
cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_mfma && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_mfma -- python3 $GRAFT_REPO_ROOT/tools/time_mfma.py > $GRAFT_REPO_ROOT/gpurun_out/prof_mfma.log 2>&1
