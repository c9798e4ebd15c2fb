python -m pytest tests/test_gpu_ffn.py tests/test_layers_golden.py tests/test_models.py -m gpu -x -q > gpurun_out/tf.log 2>&1 ; tail -25 gpurun_out/tf.log
python tools/bench_ffn.py > gpurun_out/ffn.json 2> gpurun_out/ffn.err; tail -3 gpurun_out/ffn.err; cat gpurun_out/ffn.json
bash tools/prof_ffn.sh
