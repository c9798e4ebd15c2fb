# HBM traffic per kernel launch, as MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE --pmc passes (they do not fit one pass), kernel-trace only.
cd /tmp && export TMPDIR=/tmp
OPS=${1:-sddmm,spmm_n,transpose,lookup,cdist,softmax}
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch $GRAFT_REPO_ROOT/gpurun_out/pmc_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -- python3 $GRAFT_REPO_ROOT/tools/prof_ops.py $OPS 3 > $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -- python3 $GRAFT_REPO_ROOT/tools/prof_ops.py $OPS 3 > $GRAFT_REPO_ROOT/gpurun_out/pmc_write.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/pmc_write.log
