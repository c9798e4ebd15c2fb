cd /tmp && export TMPDIR=/tmp
OPS=${1:-transpose,lookup}
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc1 -- python3 $GRAFT_REPO_ROOT/tools/prof_ops.py $OPS 2 > $GRAFT_REPO_ROOT/gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2 -- python3 $GRAFT_REPO_ROOT/tools/prof_ops.py $OPS 2 > $GRAFT_REPO_ROOT/gpurun_out/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY TCC_EA0_WRREQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc3 -- python3 $GRAFT_REPO_ROOT/tools/prof_ops.py $OPS 2 > $GRAFT_REPO_ROOT/gpurun_out/pmc3.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/pmc3.log
