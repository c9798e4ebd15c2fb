python tools/bench_block.py > gpurun_out/block.json 2> gpurun_out/block.err; tail -2 gpurun_out/block.err; cat gpurun_out/block.json
timeout -k 10 600 python tools/bench_model.py > gpurun_out/model.json 2> gpurun_out/model.err; tail -4 gpurun_out/model.err
