"""Per-cycle kernel times and idle time from a rocprofv3 kernel trace of scratch/cycle_trace.py (last 10 cycles)."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("mfmg::(anonymous namespace)::", "").replace("mfmg::vec::(anonymous namespace)::", "vec::"))
ncyc = 10
# one restriction launch per cycle and nothing after the last cycle: the period is the distance between the last two
marks = [i for i, r in enumerate(rows) if "residual_restriction" in r["Kernel_Name"] or "sr_restrict" in r["Kernel_Name"]]
period = marks[-1] - marks[-2]
assert all(marks[-k] - marks[-k - 1] == period for k in range(1, ncyc)), "the last cycles do not repeat"
first = len(rows) - period * ncyc
tail = rows[first:]
span = int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail)
print(f"kernels/cycle {len(tail)/ncyc:.1f}  span {span/ncyc/1e3:.1f} us/cycle  busy {busy/ncyc/1e3:.1f}  idle {(span-busy)/ncyc/1e3:.1f}")
acc = collections.OrderedDict()
for r in tail:
    k = name(r); a = acc.setdefault(k, [0, 0]); a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{t/ncyc/1e3:9.1f} us/cycle {c/ncyc:5.1f} x {t/c/1e3:8.1f} us  {k[:110]}")
if len(sys.argv) > 2:   # sequence of one cycle
    n = len(tail) // ncyc
    for r in tail[-n:]:
        print(f"{(int(r['Start_Timestamp'])-int(tail[-n]['Start_Timestamp']))/1e3:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}  {name(r)[:90]}  grid {r.get('Grid_Size_X','')}")
