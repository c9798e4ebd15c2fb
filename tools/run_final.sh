# end-of-round evidence: tests, bench line, rocprofv3 stats of the same command, PMC traffic
python -m pytest tests -m gpu -x -q > gpurun_out/final_tests.log 2>&1 ; tail -3 gpurun_out/final_tests.log
python __graft_entry__.py smoke > gpurun_out/final_smoke.log 2>&1; tail -1 gpurun_out/final_smoke.log
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; tail -c 200 gpurun_out/final_bench.err
python bench.py --trigger --no-cpu --no-dense > gpurun_out/final_bench_trigger.json 2>/dev/null
python tools/bench_block.py > gpurun_out/final_block.json 2>/dev/null
python tools/bench_ffn.py > gpurun_out/final_ffn.json 2>/dev/null
D_MODEL=2048 N_HEADS=32 D_FF=8192 SEQ=2048 BATCH=2 python tools/bench_block.py > gpurun_out/final_block_opt1b3.json 2>/dev/null
FAMILY=llama D_MODEL=4096 N_HEADS=32 D_FF=11008 SEQ=2048 BATCH=1 python tools/bench_block.py > gpurun_out/final_block_llama7b.json 2>/dev/null
timeout -k 10 600 python tools/bench_model.py > gpurun_out/final_model.json 2>/dev/null
python tools/bench_long.py > gpurun_out/final_long.json 2>/dev/null
python tools/time_mfma.py > gpurun_out/final_mfma_ops.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/final_prof $GRAFT_REPO_ROOT/gpurun_out/final_prof_trigger
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final_prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-dense > $GRAFT_REPO_ROOT/gpurun_out/final_prof.log 2>&1
cd $GRAFT_REPO_ROOT && bash tools/pmc_traffic.sh sddmm,spmm_n,transpose,lookup,cdist,softmax,pq_loss,fused,mfma > gpurun_out/final_traffic.log 2>&1
echo done
