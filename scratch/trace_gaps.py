"""Largest gaps between consecutive kernels of a rocprofv3 kernel trace (csv), with the kernels around them."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("mfmg::(anonymous namespace)::", ""))[:70]
last = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -2000:]
gaps = []
for i in range(1, len(last)):
    g = int(last[i]["Start_Timestamp"]) - int(last[i - 1]["End_Timestamp"])
    gaps.append((g, i))
for g, i in sorted(gaps, reverse=True)[:8]:
    print(f"gap {g/1e3:10.1f} us between [{name(last[i-1])}] and [{name(last[i])}]  (kernel {i} of the last {len(last)})")
