import sys, csv, glob, collections
tag = sys.argv[1]; last = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for f in sorted(glob.glob(f'gpurun_out/pmc_{tag}/*/*counter_collection.csv')):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'mf_laplace' not in r['Kernel_Name']:
            continue
        per[r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    for k, v in per.items():
        # several rows per dispatch (one per XCD/instance) are summed; keep the last `last` dispatches
        acc = collections.defaultdict(float)
        for d, val in v:
            acc[d] += val
        ds = sorted(acc)[-last:]
        print(f"{k:28s} dispatches {len(ds):3d}  per launch {sum(acc[d] for d in ds)/len(ds):.5g}")
