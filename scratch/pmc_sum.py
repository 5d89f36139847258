import sys, csv, glob, collections
tag = sys.argv[1]
for f in sorted(glob.glob(f'gpurun_out/pmc_{tag}/*/*counter_collection.csv')):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if 'mf_laplace' not in r['Kernel_Name']:
            continue
        k = r['Counter_Name']
        acc[k][0] += 1
        acc[k][1] += float(r['Counter_Value'])
    for k, (n, v) in acc.items():
        print(f"{k:28s} launches {n:3d}  per launch {v/n:.4g}")
