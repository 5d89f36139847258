"""Counters per launch for the kernels matching a substring (and an optional grid size) from scratch/pmc_cycle.sh output."""
import sys, csv, glob, collections
tag, pat = sys.argv[1], sys.argv[2]
grid = sys.argv[3] if len(sys.argv) > 3 else None
for f in sorted(glob.glob(f'gpurun_out/pmcc_{tag}/*/*counter_collection.csv')):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if pat not in r['Kernel_Name'] or (grid and r.get('Grid_Size', r.get('Grid_Size_X', '')) != grid):
            continue
        per[r['Counter_Name']][int(r['Dispatch_Id'])] += float(r['Counter_Value'])
    for k, acc in per.items():
        ds = sorted(acc)[-4:]
        print(f"{k:30s} dispatches {len(acc):3d}  per launch {sum(acc[d] for d in ds)/len(ds):.5g}")
