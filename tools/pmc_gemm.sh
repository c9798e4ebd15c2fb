cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_gemm
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_gemm -- python3 $GRAFT_REPO_ROOT/tools/prof_ffn.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_gemm.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/pmc_gemm.log
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_gemm/*
