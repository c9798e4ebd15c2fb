# SQ / LDS / L2 counters of the grouped GEMM alone (tools/bench_gemm.py), three passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pg1 $R/gpurun_out/pg2 $R/gpurun_out/pg3
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pg1 -- python3 $R/tools/bench_gemm.py > $R/gpurun_out/pg1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/pg2 -- python3 $R/tools/bench_gemm.py > $R/gpurun_out/pg2.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pg3 -- python3 $R/tools/bench_gemm.py > $R/gpurun_out/pg3.log 2>&1
tail -1 $R/gpurun_out/pg3.log
