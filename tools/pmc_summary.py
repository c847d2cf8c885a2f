import csv, glob, sys
for d in sorted(glob.glob('gpurun_out/pmc_sq*')):
    fs = glob.glob(f'{d}/*/*counter_collection.csv')
    if not fs: continue
    agg = {}
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].split('(')[0].replace('void mktd::', '')[:40]
        if any(s in k for s in sys.argv[1:] or ['blindrotate']):
            agg.setdefault((k, r['Counter_Name']), []).append(float(r['Counter_Value']))
    for (k, c), v in sorted(agg.items()):
        print(d.split('/')[-1], k, c, '%.4g' % (sum(v) / len(v)))
