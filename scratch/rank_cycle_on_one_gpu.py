"""What the PARTITION costs a rank of a box run, measured on one GPU.

Eight (or gx gy gz) ranks are set up as threads of this process, each with its own context and hierarchy, messages handed over
through in-memory mailboxes behind the host transport (the harness of tests/test_box_threads.py): the setup needs the real
neighbours.  Then ONE rank switches to the reflecting transport (include/mfmg_hip.h: every message it sends is copied back on
the device as the message it would have received) and runs its V-cycles alone: every kernel, packing and unpacking kernel,
stream fork and join, the all-gather of the gathered level and its replicated work -- with a wire that costs nothing.  The
other ranks idle meanwhile.  The same cycle on one rank of the same mesh size is timed beside it.

usage: rank_cycle_on_one_gpu.py [cells_per_rank] [gx,gy,gz] [rank]        (default 128 2,2,2 0)
       env TRACE=1: only the timed rank's cycles run after the marker (for rocprofv3 --kernel-trace)
           LOW_GHOST=2|4: ghost cell layers towards a lower neighbour (4, the bench's default: the whole smoother one sweep)"""
import os, sys, threading, time, json
# eight ranks as threads of one process: with the default OpenMP team per rank (16 threads each, spinning between parallel
# regions) the launching threads starve -- 4.2 ms per cycle where the same run with two threads per rank shows 2.0
os.environ.setdefault("OMP_NUM_THREADS", "2")
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mfmg_amd as M
from test_box_threads import Mailboxes

per = int(sys.argv[1]) if len(sys.argv) > 1 else 128
grid = tuple(int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "2,2,2").split(","))
timed_rank = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n_ranks = grid[0] * grid[1] * grid[2]
cells = tuple(per * g for g in grid)
length = tuple(float(g) for g in grid)
params = {"eigensolver": {"number of eigenvectors": 2}, "agglomeration": {"partitioner": "block", "nx": 2, "ny": 2, "nz": 2},
          "smoother": {"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0, "n_smoothing_steps": 1},
          "solver": {"type": "amg", "amg": {"smoother_degree": 1, "smoothing_range": 4.0, "n_cycles": 1, "aggregate_block": 2,
                                            "pre_smoothing_levels": 0}},
          "is preconditioner": False, "max levels": 2}
material = os.environ.get("MATERIAL", "constant")
if os.environ.get("AMG_REPLICATE_ROWS"):      # levels with fewer global rows are gathered and solved redundantly (default 200000)
    params["solver"]["amg"]["replicate_rows"] = int(os.environ["AMG_REPLICATE_ROWS"])

host_enqueue_ms = []   # (appended by cycles(): ms the host spends enqueueing one cycle; if it equals the cycle time the run is launch-bound)


def cycles(ctx, h, n, steps=10, warmup=3):
    """ms per cycle: median over five blocks of `steps` / 2 cycles, the collector off meanwhile (this process holds the Python
    objects of eight ranks: a generation-2 collection in the middle of a block cost 20-90 ms in the first runs of round 4)."""
    import gc
    x = torch.rand(n, dtype=torch.float64, device="cuda"); b = torch.zeros_like(x)
    for _ in range(warmup):
        h.apply(b, x)
    gc.collect()
    gc.disable()
    try:
        blocks = []
        per = max(steps // 2, 1)
        for _ in range(5):
            ctx.synchronize(); torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(per):
                h.apply(b, x)
            t_host = time.perf_counter() - t          # every launch of the block is enqueued: what the HOST needs per cycle
            ctx.synchronize(); torch.cuda.synchronize()
            blocks.append((time.perf_counter() - t) / per * 1e3)
            host_enqueue_ms.append(t_host / per * 1e3)
    finally:
        gc.enable()
    return sorted(blocks)[len(blocks) // 2]

# ---- what one grouped RCCL send/recv costs its stream before a byte crosses a wire: loop-back on this GPU
loopback_us = None
try:
    _ctx = M.Context()
    _tr = M.HaloTransport(_ctx, M.SlabPartition((8, 8, 8), 0, 1), transport="rccl")
    loopback_us = {"32768 doubles": _tr.loopback_time(32768, 50), "2048 doubles": _tr.loopback_time(2048, 50)}
    del _tr, _ctx
except Exception as e:  # noqa: BLE001
    loopback_us = {"error": str(e)[:200]}
delays = [float(v) for v in os.environ.get("DELAY_US", "").split(",") if v] or \
    ([round(loopback_us["2048 doubles"], 1)] if "2048 doubles" in loopback_us else [25.0])

if os.environ.get("TRACE") == "1":  # (kernel trace: the LAST cycles of the process must be the free-wire ones)
    delays = []

# ---- one rank of the same size, no partition
ctx1 = M.Context()
p1 = M.LaplaceProblem((per,) * 3, material, device="cuda")
h1 = M.Hierarchy(ctx1, "HipMatrixFreeMeshEvaluator", p1, params)
ms_single = cycles(ctx1, h1, h1.level_size(0))
del h1, p1
torch.cuda.empty_cache()

mb = Mailboxes(n_ranks)
errors = [None] * n_ranks
result = {}
ready = threading.Barrier(n_ranks, timeout=3600)

def worker(rank):
    try:
        torch.cuda.set_device(0)
        part = M.BoxPartition(cells, rank, grid, length=length, low_ghost_cells=int(os.environ.get("LOW_GHOST", "4")))
        ctx = M.Context()
        tr = M.HaloTransport(ctx, part, callbacks=mb.callbacks(rank))
        t0 = time.perf_counter()
        h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", part.local_problem(material, "cuda"), params)
        ctx.synchronize()
        setup_s = time.perf_counter() - t0
        ready.wait()
        if rank == timed_rank:
            tr.reflect()
            e0, v0, o0 = tr.n_exchanges(), tr.exchange_volume(), tr.n_overlapped()
            if os.environ.get("DEBUG_PHASES") == "1":
                # wall clock of the pieces of a cycle, each synchronised: where a rank stalls
                nf, nc = h.level_size(0), h.level_size(1)
                xx = torch.rand(nf, dtype=torch.float64, device="cuda"); bb = torch.rand_like(xx); yy = torch.empty_like(xx)
                xc = torch.rand(nc, dtype=torch.float64, device="cuda"); yc = torch.empty_like(xc)
                def tm(name, f, reps=10):
                    f(); ctx.synchronize()
                    t = time.perf_counter()
                    for _ in range(reps):
                        f()
                    ctx.synchronize()
                    print(f"[phase] {name:28s} {(time.perf_counter() - t) / reps * 1e3:8.3f} ms", flush=True)
                tm("smoother_apply", lambda: h.smoother_apply(0, bb, xx))
                tm("operator_apply", lambda: h.operator_apply(0, xx, yy))
                tm("restrict_residual", lambda: h.restrict_residual(xx, bb, yc))
                tm("coarse_apply", lambda: h.coarse_apply(xc, yc))
                tm("prolongation", lambda: h.restrictor_apply(1, xc, yy, 1))
                tm("whole cycle", lambda: h.apply(bb, xx))
                zz = torch.zeros_like(xx)
                tm("whole cycle, b = 0", lambda: h.apply(zz, xx))
                x2 = torch.rand(nf, dtype=torch.float64, device="cuda")
                each = []
                for _ in range(14):
                    ctx.synchronize(); t = time.perf_counter()
                    h.apply(zz, x2)
                    ctx.synchronize(); each.append(round((time.perf_counter() - t) * 1e3, 3))
                print("[phase] per-cycle ms, fresh x, b = 0:", each, "max |x|", float(x2.abs().max()), flush=True)
            del host_enqueue_ms[:]
            ms = cycles(ctx, h, h.level_size(0))
            result["host_enqueue_ms_per_cycle"] = sorted(host_enqueue_ms)[len(host_enqueue_ms) // 2]
            n_cyc = 3 + 5 * 5
            result.update({"rank": rank, "grid": list(grid), "cells_per_rank": per, "local_cells": list(part.local_cells),
                           "ms_per_cycle_rank_alone_reflecting": ms, "ms_per_cycle_one_rank_same_size": ms_single,
                           "ratio": ms / ms_single, "exchanges_per_cycle": (tr.n_exchanges() - e0) / n_cyc,
                           "overlapped_per_cycle": (tr.n_overlapped() - o0) / n_cyc,
                           "mb_sent_per_cycle": (tr.exchange_volume() - v0) * 8e-6 / n_cyc, "setup_seconds_in_threads": setup_s,
                           "gathered_from_rows": h.coarse_amg_gather_rows(), "levels": h.coarse_amg_shapes(),
                           "smoother_sweep_terms": list(h.smoother_sweep_terms()), "low_ghost_cells": part.low_ghost_cells, "rccl_loopback_us_per_group": loopback_us})
            # ... and with a price on the wire: every grouped send/recv and collective holds its stream that long
            with_delay = {}
            for d_us in delays:
                tr.reflect(d_us)
                with_delay[f"{d_us:g} us per group"] = cycles(ctx, h, h.level_size(0))
            result["ms_per_cycle_rank_alone_with_wire_latency"] = with_delay
        ready.wait()
        del h
    except BaseException as e:  # noqa: BLE001
        errors[rank] = e
        mb.barrier.abort(); ready.abort()
        for p in range(n_ranks):
            mb.q[(rank, p)].put(np.zeros(0))

threads = [threading.Thread(target=worker, args=(r,)) for r in range(n_ranks)]
for t in threads:
    t.start()
for t in threads:
    t.join()
first = next((e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)), None) or \
    next((e for e in errors if e is not None), None)
if first is not None:
    raise first
print(json.dumps(result))
