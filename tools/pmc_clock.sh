cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_clock
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_clock -- python3 $GRAFT_REPO_ROOT/tools/bench_gemm.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_clock.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/pmc_clock.log
