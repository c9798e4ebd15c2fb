python -m pytest tests -m gpu -x -q > gpurun_out/t5.log 2>&1 ; tail -4 gpurun_out/t5.log
python tools/bench_ffn.py > gpurun_out/ffn.json 2> gpurun_out/ffn.err; tail -2 gpurun_out/ffn.err; cat gpurun_out/ffn.json
python tools/bench_block.py > gpurun_out/block.json 2> gpurun_out/block.err; tail -2 gpurun_out/block.err; cat gpurun_out/block.json
