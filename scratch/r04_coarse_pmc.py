# usage (on the GPU box): python3 scratch/r04_coarse_pmc.py <tag>
# Memory-side counters of the three large kernels of the coarse part inside the V-cycle (257^3 DoFs, material constant): the
# one-pass residual restriction, the first coarse operator's class kernel, the prolongation.  One `rocprofv3 --kernel-trace --pmc
# <group>` pass per small group around scratch/cycle_trace.py, the program directly behind `--`; this driver does not touch the GPU.
import collections
import csv
import glob
import os
import re
import subprocess
import sys

tag = sys.argv[1]
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join(R, "gpurun_out", f"r04cpmc_{tag}")
os.makedirs(out, exist_ok=True)
os.chdir("/tmp")
os.environ["TMPDIR"] = "/tmp"
avail = set(re.findall(r"\b([A-Z][A-Za-z0-9_]{3,})\b", subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True).stdout))
wish = [
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAVES", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU"],
    ["SQ_INST_LEVEL_VMEM", "SQ_LEVEL_WAVES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY"],
    ["TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum"],
    ["TCP_TCC_READ_REQ_LATENCY_sum", "TCP_PENDING_STALL_CYCLES_sum"],
    ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_LEVEL_sum"],
    ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_WRREQ_LEVEL_sum"],
    ["TA_TA_BUSY_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum"],
    ["GRBM_GUI_ACTIVE"],
]
kernels = {"residual_restriction": None, "sr_prolong": None, "bdia_class_node": "max"}  # (class kernel: the launch with the largest grid = A_c)
summary = {k: collections.OrderedDict() for k in kernels}
prog = ["python3", os.path.join(R, "scratch", "cycle_trace.py"), "256", "constant"]
for group in wish:
    have = [c for c in group if c in avail]
    if not have:
        print("not offered:", group, flush=True)
        continue
    d = os.path.join(out, have[0])
    res = subprocess.run(["timeout", "-k", "10", "240", "rocprofv3", "--kernel-trace", "--pmc"] + have + ["--output-format", "csv", "-d", d, "-o", "p", "--"] + prog,
                         capture_output=True, text=True)
    print("pass", have, "rc", res.returncode, flush=True)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        for kname, sel in kernels.items():
            mine = [r for r in rows if kname in r["Kernel_Name"]]
            if sel == "max" and mine:
                g = max(int(r.get("Grid_Size", r.get("Grid_Size_X", "0"))) for r in mine)
                mine = [r for r in mine if int(r.get("Grid_Size", r.get("Grid_Size_X", "0"))) == g]
            per = collections.defaultdict(lambda: collections.defaultdict(float))
            for r in mine:
                per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
            for c, acc in per.items():
                ds = sorted(acc)[-4:]
                summary[kname][c] = sum(acc[x] for x in ds) / len(ds)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        for kname, sel in kernels.items():
            mine = [r for r in rows if kname in r["Kernel_Name"]]
            if sel == "max" and mine:
                g = max(int(r.get("Grid_Size", r.get("Grid_Size_X", "0"))) for r in mine)
                mine = [r for r in mine if int(r.get("Grid_Size", r.get("Grid_Size_X", "0"))) == g]
            dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in mine][-4:]
            if dur:
                summary[kname]["_kernel_ns_" + have[0]] = sum(dur) / len(dur)
        os.remove(f)
with open(os.path.join(R, "gpurun_out", f"r04_n_coarse_kernels_pmc_{tag}.txt"), "w") as fo:
    for kname, acc in summary.items():
        print(f"== {kname} (mean of the last 4 launches, counters summed over their instances)", file=fo)
        for c, v in acc.items():
            print(f"{c:40s} {v:.6g}", file=fo)
        g = acc.get
        if g("TCC_EA0_RDREQ_sum") and g("TCC_EA0_RDREQ_LEVEL_sum") and g("GRBM_GUI_ACTIVE"):
            ns = g("_kernel_ns_TCC_EA0_RDREQ_sum") or 0.
            print(f"-> fabric reads {g('TCC_EA0_RDREQ_sum'):.4g} (32 B: {g('TCC_EA0_RDREQ_32B_sum', 0.):.3g}); cycles per read in the L2's queue "
                  f"{g('TCC_EA0_RDREQ_LEVEL_sum') / g('TCC_EA0_RDREQ_sum'):.0f}; kernel {ns / 1e3:.1f} us in that pass", file=fo)
print(open(os.path.join(R, "gpurun_out", f"r04_n_coarse_kernels_pmc_{tag}.txt")).read())
