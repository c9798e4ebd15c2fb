python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/bench3.json 2> gpurun_out/bench3.err; tail -c 300 gpurun_out/bench3.err
bash tools/run_trig.sh 2>&1 | grep -E "^trigger|^plain|^trig "
cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof3 && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu --no-dense > $GRAFT_REPO_ROOT/gpurun_out/prof3.log 2>&1
