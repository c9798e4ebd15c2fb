python -m pytest tests/test_models.py tests/test_gpu_pq_loss.py -m gpu -x -q > gpurun_out/tm.log 2>&1 ; tail -5 gpurun_out/tm.log
timeout -k 10 600 python tools/bench_model.py > gpurun_out/model.json 2> gpurun_out/model.err; tail -5 gpurun_out/model.err
