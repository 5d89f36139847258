"""HBM bytes per launch of every kernel of the cycle from scratch/pmc_cycle_mem.sh (FETCH_SIZE and WRITE_SIZE are in
units of 32 B... see MI355X_MICROARCH.md: on gfx950 FETCH_SIZE is reported in kB and needs the factor 2)."""
import sys, csv, glob, collections, re
tag = sys.argv[1]
name = lambda k: re.sub(r"\(.*", "", k.replace("void ", "").replace("mfmg::(anonymous namespace)::", "").replace("mfmg::vec::(anonymous namespace)::", "vec::"))
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set); dur = collections.defaultdict(float)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmcc_{tag}/{c}/*counter_collection.csv"):
        rows = list(csv.DictReader(open(f)))
        last = max(int(r["Dispatch_Id"]) for r in rows)
        marks = sorted({int(r["Dispatch_Id"]) for r in rows if "residual_restriction" in r["Kernel_Name"] or "sr_restrict" in r["Kernel_Name"]})
        period = marks[-1] - marks[-2]                      # one restriction launch per cycle
        for r in rows:
            if int(r["Dispatch_Id"]) <= last - period * 5:  # the last 5 cycles
                continue
            k = (name(r["Kernel_Name"]), r["Grid_Size"])
            tot[k][c] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
            if c == "FETCH_SIZE":
                dur[k] += 0
print(f"{'kernel':58s} {'grid':>9s} {'n':>3s} {'read MB':>9s} {'write MB':>9s}")
for k in sorted(tot, key=lambda k: -tot[k]["FETCH_SIZE"]):
    n = max(len(cnt[k]), 1)
    rd = tot[k]["FETCH_SIZE"] * 2 * 1024 / n / 1e6      # kB, x2 on gfx950
    wr = tot[k]["WRITE_SIZE"] * 1024 / n / 1e6
    print(f"{k[0][:58]:58s} {k[1]:>9s} {n:3d} {rd:9.1f} {wr:9.1f}")
