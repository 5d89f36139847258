"""Cost of one grouped RCCL send/recv as the halo exchange issues it, measured on ONE GPU through the library's loop-back
(ncclGroupStart; ncclSend to self; ncclRecv from self; ncclGroupEnd on the library's stream): `mfmg_hip_context_transport_selftest`
runs a loop-back, an all-gather and two all-reduces, each followed by a stream synchronisation, so the figure printed here is
an UPPER bound for the launch-and-complete latency of four small RCCL operations -- no wire is involved.
usage: rccl_latency.py [n_doubles]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mfmg_amd as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
ctx = M.Context()
part = M.SlabPartition((8, 8, 8), 0, 1)
tr = M.HaloTransport(ctx, part, transport="rccl")
assert tr.name() == "rccl"
for _ in range(5):
    tr.selftest(n)
torch.cuda.synchronize()
for m in (n, 8 * n):
    t = time.perf_counter()
    reps = 50
    for _ in range(reps):
        tr.selftest(m)
    torch.cuda.synchronize()
    print(f"selftest({m} doubles = {m * 8 / 1024:.0f} KiB): {(time.perf_counter() - t) / reps * 1e6:.1f} us for loop-back + all-gather + 2 all-reduces, each synchronised")
