import csv, glob, collections, os, sys
for d in sys.argv[1:]:
    f = glob.glob(f'/root/repo/gpurun_out/{d}/*/*counter_collection.csv')
    if not f: print('no file', d); continue
    f.sort(key=os.path.getmtime)                 # gpurun merges runs: newest pass
    rows = list(csv.DictReader(open(f[-1])))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        name = r['Kernel_Name'][:48]
        if 'spt::' not in name: continue
        agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(k)
        for c, vals in v.items():
            print('   ', c, sum(vals)/len(vals))
