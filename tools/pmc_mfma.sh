cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_mfma $GRAFT_REPO_ROOT/gpurun_out/pmc_mfma2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_mfma -- python3 $GRAFT_REPO_ROOT/tools/time_mfma.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_mfma.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_mfma2 -- python3 $GRAFT_REPO_ROOT/tools/time_mfma.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_mfma2.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/pmc_mfma2.log
