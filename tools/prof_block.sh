# rocprofv3 kernel stats of the sparse TransformerBlock step (bench_block.py, TUNINGS=sparse)
export TUNINGS=sparse
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_block
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_block -- python3 $GRAFT_REPO_ROOT/tools/bench_block.py > $GRAFT_REPO_ROOT/gpurun_out/prof_block.log 2>&1
cd $GRAFT_REPO_ROOT && python - <<'P'
import csv, glob
f = sorted(glob.glob('gpurun_out/prof_block/*/*_kernel_stats.csv'))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
calls = sum(int(r['Calls']) for r in rows)
print('total kernel ms', tot / 1e6, 'calls', calls, 'per step (15 steps): ms', tot / 15e6, 'launches', calls / 15)
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs'])/15e3:9.1f} us/step {int(r['Calls'])/15:6.1f} x  {r['Name'][:110]}")
P
