python -m pytest tests -m gpu -x -q > gpurun_out/t6.log 2>&1 ; tail -8 gpurun_out/t6.log
python tools/bench_ffn.py > gpurun_out/ffn1.json 2> gpurun_out/ffn1.err; tail -c 400 gpurun_out/ffn1.err; cat gpurun_out/ffn1.json
