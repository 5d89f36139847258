"""Kernel times per cycle of the LAST ten V-cycles in a rocprofv3 kernel trace (csv), cycles delimited by their
residual_restriction_kernel launch: for scratch/rank_cycle_on_one_gpu.py, whose timed rank runs alone at the end of the process.
usage: rank_cycle_trace_sum.py kernel_trace.csv [seq]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("mfmg::(anonymous namespace)::", "").replace("mfmg::vec::(anonymous namespace)::", "vec::"))
rr = [i for i, r in enumerate(rows) if "residual_restriction" in r["Kernel_Name"]]
ncyc = 10
tail = rows[rr[-ncyc - 1]: rr[-1]]
span = int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail)
print(f"launches/cycle {len(tail)/ncyc:.1f}  span {span/ncyc/1e3:.1f} us/cycle  sum of kernel times {busy/ncyc/1e3:.1f} (kernels of two streams overlap)")
acc = collections.OrderedDict()
for r in tail:
    k = name(r); a = acc.setdefault(k, [0, 0]); a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{t/ncyc/1e3:9.1f} us/cycle {c/ncyc:5.1f} x {t/c/1e3:8.1f} us  {k[:110]}")
if len(sys.argv) > 2:
    one = rows[rr[-2]: rr[-1]]
    t0 = int(one[0]["Start_Timestamp"])
    for r in one:
        print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}  {name(r)[:90]}  grid {r.get('Grid_Size_X','')}")
