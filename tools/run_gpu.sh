python -m pytest tests -m gpu -x -q > gpurun_out/t5.log 2>&1 ; tail -15 gpurun_out/t5.log
python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/bench3.json 2> gpurun_out/bench3.err; tail -c 300 gpurun_out/bench3.err
python bench.py --steps 20 --warmup 5 --no-cpu --trigger > gpurun_out/bench3t.json 2> gpurun_out/bench3t.err; tail -c 300 gpurun_out/bench3t.err
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu --no-dense --trigger > $GRAFT_REPO_ROOT/gpurun_out/prof3.log 2>&1
