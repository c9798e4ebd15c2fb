import csv,glob,collections,sys
for tag in sys.argv[1:]:
    f=sorted(glob.glob('gpurun_out/prof_%s/*/*_kernel_trace.csv'%tag))[-1]
    c=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'grouped_gemm' in r['Kernel_Name']:
            c[int(r['Grid_Size_X'])//256].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    print(tag, {k:round(sorted(v)[len(v)//2],1) for k,v in sorted(c.items())})
