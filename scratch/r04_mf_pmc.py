# usage (on the GPU box): python3 scratch/r04_mf_pmc.py <tag> <n> <material> [ty tz waves]
# memory-side counters of the operator kernel, one --pmc pass per small group (kernel-trace only, the
# program directly behind `--`).  This driver does not touch the GPU itself: it lists the counters gfx950 offers
# (rocprofv3 -L), keeps the ones of the wish list that exist and runs scratch/smoother_only.py once per pass.
import csv
import collections
import glob
import os
import re
import subprocess
import sys

tag, n, material = sys.argv[1], sys.argv[2], sys.argv[3]
tile = sys.argv[4:7] if len(sys.argv) >= 7 else ["0", "0", "0"]
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join(R, "gpurun_out", f"r04pmc_{tag}")
os.makedirs(out, exist_ok=True)
os.chdir("/tmp")
os.environ["TMPDIR"] = "/tmp"

avail_txt = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True).stdout
open(os.path.join(out, "counters_available.txt"), "w").write(avail_txt)
avail = set(re.findall(r"\b([A-Z][A-Za-z0-9_]{3,})\b", avail_txt))

wish = [
    # what a vector read becomes on its way to the L2
    ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INST_LEVEL_VMEM", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"],
    ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_SALU"],
    ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum"],
    ["TCP_TOTAL_ACCESSES_sum", "TCP_TOTAL_READ_sum"],
    ["TCP_TCC_READ_REQ_LATENCY_sum", "TCP_PENDING_STALL_CYCLES_sum"],
    ["TCP_GATE_EN1_sum", "TCP_GATE_EN2_sum"],
    ["TCP_TD_TCP_STALL_CYCLES_sum", "TCP_TCR_TCP_STALL_CYCLES_sum"],
    ["TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", "TCP_TA_TCP_STATE_READ_sum"],
    ["TCP_TCC_NC_READ_REQ_sum", "TCP_TCC_UC_READ_REQ_sum"],
    ["TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_TA_BUSY_sum"],
    ["TA_FLAT_READ_WAVEFRONTS_sum", "TA_BUFFER_READ_WAVEFRONTS_sum"],
    ["TA_DATA_STALLED_BY_TC_CYCLES_sum", "TA_ADDR_STALLED_BY_TD_CYCLES_sum"],
    ["TD_TD_BUSY_sum", "TD_TC_STALL_sum"],
    ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"],
    ["TCC_REQ_sum", "TCC_READ_sum", "TCC_TAG_STALL_sum"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_LEVEL_sum"],
    ["TCC_EA0_RD_UNCACHED_32B_sum", "TCC_EA0_RDREQ_DRAM_sum"],
    ["TCC_BUSY_sum", "TCC_STREAMING_REQ_sum", "TCC_PROBE_sum"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE", "GRBM_GUI_ACTIVE"],
]
# per-XCD request counts where the list offers indexed forms
per_xcd = sorted(c for c in avail if re.fullmatch(r"TCC_REQ\[\d+\]", c))

prog = ["python3", os.path.join(R, "scratch", "smoother_only.py"), n] + tile[:2] + ["2"]
if os.environ.get("SWEEP_K"):  # the multi-term sweep instead: tile = nw ty tz
    prog = ["python3", os.path.join(R, "scratch", "sweep_only.py"), n, os.environ["SWEEP_K"]] + tile[:3] + ["3"]
if os.environ.get("PASSES"):
    wish = wish[:int(os.environ["PASSES"])]
env = dict(os.environ, MATERIAL=material, WAVES=tile[2])
summary = collections.OrderedDict()
missing = []
for group in wish:
    have = [c for c in group if c in avail]
    missing += [c for c in group if c not in avail]
    if not have:
        continue
    d = os.path.join(out, have[0])
    cmd = ["timeout", "-k", "10", "240", "rocprofv3", "--kernel-trace", "--pmc"] + have + ["--output-format", "csv", "-d", d, "-o", "p", "--"] + prog
    res = subprocess.run(cmd, capture_output=True, text=True, env=env)
    open(os.path.join(out, have[0] + ".log"), "w").write(res.stdout[-4000:] + res.stderr[-4000:])
    print("pass", have, "rc", res.returncode, flush=True)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if "mf_laplace" not in r["Kernel_Name"] and "mf_cheb" not in r["Kernel_Name"]:
                continue
            per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
        for k, acc in per.items():
            ds = sorted(acc)[-6:]
            summary[k] = sum(acc[x] for x in ds) / len(ds)
    # kernel durations of this pass (ns), for the record
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "mf_laplace" in r["Kernel_Name"] or "mf_cheb" in r["Kernel_Name"]]
        if dur:
            summary.setdefault("_kernel_ns_" + have[0], sum(dur[-6:]) / len(dur[-6:]))

with open(os.path.join(out, "summary.txt"), "w") as fh:
    fh.write(f"== {n}^3 DoFs, material {material}, tile {tile} (mean of the last 6 dispatches of the operator kernel, summed over instances)\n")
    for k, v in summary.items():
        fh.write(f"{k:44s} {v:.6g}\n")
    fh.write("not offered on this device: " + " ".join(missing) + "\n")
    fh.write("indexed TCC_REQ forms: " + " ".join(per_xcd) + "\n")
print(open(os.path.join(out, "summary.txt")).read())
