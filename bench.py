#!/usr/bin/env python3
"""bench.py -- V-cycle throughput of the MI355X-native mfmg hot path.

One "step" = one `Hierarchy::apply` (V-cycle: Chebyshev(3) pre-smooth, residual, restriction, coarse
solve = one V-cycle of the smoothed-aggregation hierarchy below the first coarse level -- V(0,1), i.e. post-smoothing
only, on its levels by default (`--amg-pre-levels`; the symmetric V(1,1) cycle is measured beside it) --, prolongation,
Chebyshev(3) post-smooth) on the matrix-free Q1 Laplace of a synthetic 3-D hyper-cube, FP64, inputs
resident in HBM.  The `cpu_baseline` leg runs the same cycle in the oracle's C++ port and compares its iterates with the
GPU's (`parity_vs_gpu`).  Metric (BASELINE.json): fine-DoFs/sec per V-cycle.  One process per GPU; rank 0
prints ONE JSON line.

Byte accounting of the roofline block: `achieved` = ALGORITHMIC bytes of the smoother terms one launch of the
dominant kernel performs -- per term what the data layout requires of a launch of that term (x, out, one id, the
coefficients, the epilogue operands; halo re-reads excluded); the multi-term sweep performs three terms per launch and is
priced on the three, with the bytes the fused form itself must move printed beside it -- / the average launch time from HIP
events on the launch stream; `traffic` = HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/
for exactly this workload and tile (null otherwise); SURVEY.md 8(d)'s own figures (112 B/DoF per operator application, 144
per fused smoother term, 432 per Chebyshev(3) apply) are printed under `survey_8d_*`.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cells", type=int, default=256,
                    help="cells per direction PER GPU (deal.II refine_global(r): 2^r cells, 2^r + 1 DoFs)")
    ap.add_argument("--degree", type=int, default=3, help="Chebyshev degree of the smoother")
    ap.add_argument("--coarse", default="amg", choices=["amg", "pcg"],
                    help="coarse 'solve': one V-cycle of the smoothed-aggregation hierarchy, or Jacobi-PCG steps")
    ap.add_argument("--coarse-iters", type=int, default=10, help="Jacobi-PCG steps when --coarse pcg")
    ap.add_argument("--material", default="constant")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cycles", type=int, default=5, help="timed V-cycles of the CPU baseline sample")
    ap.add_argument("--no-smoother-512", action="store_true",
                    help="skip the extra fine-level smoother measurement at 512^3 (north_star target config)")
    ap.add_argument("--no-vcycle-513", action="store_true",
                    help="skip the one-GPU V-cycle on the global problem of configs[3] (512^3 cells; ~40 s with its setup)")
    ap.add_argument("--amg-block", type=int, default=2, help="nodes per direction of one aggregate of the coarse AMG")
    ap.add_argument("--amg-deep", type=str, default="", help="level,block: bigger geometric aggregates from that AMG level on")
    ap.add_argument("--amg-pre-levels", type=int, default=0,
                    help="levels of the coarse AMG (from its top) that pre-smooth; the ones below run V(0,1) (default 0: post-"
                         "smoothing only on every level of the aggregation hierarchy; -1: V(1,1) on all of them)")
    ap.add_argument("--amg-degree", type=int, default=1, help="Chebyshev degree of the coarse AMG smoothers")
    ap.add_argument("--amg-replicate-rows", type=int, default=0,
                    help="distributed runs: aggregation levels with at most this many global rows are gathered and solved "
                         "redundantly on every rank (0: the library's default, 200000)")
    ap.add_argument("--amg-setup", default="", choices=["", "device", "host"],
                    help="aggregation hierarchy by probing on the device or by host SpGEMM (default: the library's choice)")
    ap.add_argument("--evaluator", default="matrix_free", choices=["matrix_free", "assembled"],
                    help="fine-level operator: matrix-free (BASELINE configs[1]/[3]) or assembled CSR (configs[2])")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra single-GPU lines (configs[1] 128^3 V-cycle, configs[4] FP32 smoother apply)")
    ap.add_argument("--allow-missing-extras", action="store_true",
                    help="report an extra leg that raises as an `error` entry instead of failing the run")
    ap.add_argument("--box", type=str, default="", help="gx,gy,gz: global cells of a distributed run instead of the "
                    "weak-scaling box (rehearsals of the slab shapes of larger runs on fewer ranks)")
    ap.add_argument("--partition", default="box", choices=["box", "slab"],
                    help="N > 1: equal boxes on the grid the weak-scaling mesh grows by (1x1x2, 1x2x2, 2x2x2; SURVEY.md 8e) or slabs "
                         "along z")
    ap.add_argument("--grid", type=str, default="", help="px,py,pz: ranks along x, y, z of a box run (default: follows the mesh)")
    ap.add_argument("--low-ghost", type=int, default=4, choices=[2, 4],
                    help="N > 1: ghost cell layers of a rank's local mesh towards a lower neighbour: 4 (two agglomerates: the whole "
                         "Chebyshev(3) smoother of a rank is ONE sweep, x exchanged once and three planes deep) or 2 (rounds 2-3: "
                         "two terms per sweep + a launch for the third, two exchanges)")
    ap.add_argument("--tile", type=str, default="", help="ty,tz[,waves] override of the operator tile (default: timed choice)")
    return ap.parse_args()


def smoother_coefficients(degree, lmin, lmax):
    """(alpha_k, beta_k) of the three-term Chebyshev recurrence (deal.II PreconditionChebyshev)."""
    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    out = [(0.0, 1.0 / theta)]
    if degree >= 2 and abs(delta) >= 1e-40:
        rhok, sigma = delta / theta, theta / delta
        for _ in range(degree - 1):
            rhokp = 1.0 / (2.0 * sigma - rhok)
            out.append((rhokp * rhok, 2.0 * rhokp / delta))
            rhok = rhokp
    return out


def committed_traffic(cells, degree, compact, tile, prefix="cells", want_source=False):
    """HBM bytes per launch of the operator kernel from the PMC passes committed under profiles/ (2 x FETCH_SIZE
    + WRITE_SIZE, MI355X_MICROARCH.md): counters need rocprofv3 around the process, so this is not a live reading.
    The file is keyed on the workload AND the tile; anything else returns None."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "mf_kernel_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f)
    key = f"{prefix}{cells}_degree{degree}_{'cell_constant' if compact else 'general'}_tile{'x'.join(str(v) for v in tile)}"
    e = d.get(key)
    if want_source:
        return (e.get("traffic_bytes_per_launch"), f"{e.get('source')}; {e.get('date')}") if e else (None, None)
    return e.get("traffic_bytes_per_launch") if e else None


def operator_bytes_per_dof(word, compact, survey, ids_computed=False):
    """Bytes per fine DoF of one operator application y = A x.
    survey=True : SURVEY.md 8(d)'s indexed form -- x + y + 8 index ints + 8 coefficients (112 B in FP64); with one
                  coefficient per cell 8 indices + 1 coefficient.
    survey=False: what the chunk-record layout makes the kernel read at least -- x + y + ONE id (each slot stores
                  its own DoF id, the other seven corners are neighbours' own ids) + the coefficients (8, or 1 for a
                  cell-wise constant material).  Halo re-reads of the tiling are NOT in this figure: they are waste."""
    coef = (1 if compact else 8) * word
    # (ids_computed: the kernel computes the ids of a structured numbering instead of reading them -- not a required byte)
    return 2 * word + (32 if survey else (0 if ids_computed else 4)) + coef


def survey_8d_bytes_per_dof(n_terms, word=8):
    """SURVEY.md 8(d), whatever the layout: operator application 112 B/DoF in FP64 (x + y + 8 index ints + 8 coefficients; 72
    in FP32), + 32 (b, D^-1, x_prev, x+) per fused smoother term = 144, Chebyshev(3) apply = 3 terms = 432."""
    op = 2 * word + 8 * 4 + 8 * word
    return {"operator": op, "smoother_term": op + 4 * word, "smoother_apply": n_terms * (op + 4 * word)}


def smoother_bytes_per_dof(n_terms, word, compact, survey, dinv_stored=None, ids_computed=False):
    """Chebyshev smoother apply = n_terms fused operator launches: + b + D^-1 each, + x_prev from the second on.  With one
    coefficient per cell the layout holds no D^-1 by default (the kernel derives it from the cell coefficients): then it
    is not a required byte; `dinv_stored` says what the operator at hand does."""
    b_op = operator_bytes_per_dof(word, compact, survey, ids_computed)
    if dinv_stored is None:
        dinv_stored = not compact
    dinv = word if (survey or dinv_stored) else 0
    return (b_op + word + dinv) + (n_terms - 1) * (b_op + 2 * word + dinv)


def ms_per_decade(ms_per_cycle, contraction):
    """Time to reduce the residual by a factor of ten at the measured contraction per cycle: the figure that prices a change
    of the CYCLE (V(0,1) against V(1,1) on the aggregation levels: ADVICE r03) -- cheaper cycles that contract less do not win."""
    if not (0.0 < contraction < 1.0):
        return None
    return ms_per_cycle / (-math.log10(contraction))


def measure_vcycle_small(ctx, torch, M, cells, params, steps=10, warmup=3, material="constant",
                         evaluator="HipMatrixFreeMeshEvaluator", release_setup_matrices=False):
    """BASELINE.json configs[1]: the same V-cycle on a `cells`^3 mesh, wall clock around `steps` cycles.
    release_setup_matrices: the hierarchy frees the CSR arrays of its table-driven operators once it stands (the same cycle bit
    for bit, tests/test_gpu_hierarchy.py::test_release_setup_matrices; only the exports of those matrices are gone)."""
    if release_setup_matrices:
        params = dict(params)
        params["release setup matrices"] = True
    prob = M.LaplaceProblem((cells,) * 3, material, device="cuda")
    t_setup = time.perf_counter()
    h = M.Hierarchy(ctx, evaluator, prob, params)
    ctx.synchronize()
    t_setup = time.perf_counter() - t_setup
    n = h.level_size(0)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    x *= (prob.constrained != 1).to(torch.float64)
    b = torch.zeros(n, dtype=torch.float64, device="cuda")
    # residual norms around the run (b = 0): the contraction the cycle delivers at this size
    r = torch.empty_like(x)
    h.operator_apply(0, x, r)
    res0 = ctx.l2_norm(r)
    for _ in range(warmup):
        h.apply(b, x)
    # three blocks of `steps` cycles, the median block: a host hiccup inside ten cycles of 0.3 ms (a collection of the Python
    # objects of the legs before: 0.53 instead of 0.30 ms in one run of round 4) is not the cycle's time
    blocks = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            h.apply(b, x)
        torch.cuda.synchronize()
        blocks.append((time.perf_counter() - t0) / steps)
    dt = sorted(blocks)[1]
    h.operator_apply(0, x, r)
    res1 = ctx.l2_norm(r)
    # contraction over the first 8 cycles from a fresh start (the figure ms_per_residual_decade is priced on)
    x = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * (prob.constrained != 1).to(torch.float64)
    h.operator_apply(0, x, r)
    r8_0 = ctx.l2_norm(r)
    for _ in range(8):
        h.apply(b, x)
    h.operator_apply(0, x, r)
    contraction8 = (ctx.l2_norm(r) / r8_0) ** 0.125 if r8_0 > 0 else 0.0
    # device memory of the cycle's state: the caching allocator of torch hands back what the setup and this function no longer
    # hold (otherwise its cache -- problem arrays of the legs before, temporaries -- is counted as in use)
    del r
    torch.cuda.empty_cache()
    free_b, total_b = torch.cuda.mem_get_info()
    torch_gb = torch.cuda.memory_allocated() / 1e9
    kind = "matrix-free" if evaluator == "HipMatrixFreeMeshEvaluator" else "assembled CSR fine operator"
    return {"workload": f"{cells}^3 cells = {cells + 1}^3 DoFs, {kind}, material {material}, Chebyshev(3), same "
                        f"hierarchy parameters",
            "n_dofs": n, "ms_per_step": dt * 1e3, "value": n / dt, "unit": "DoF/s", "setup_seconds": t_setup,
            "ms_per_step_blocks": [v * 1e3 for v in blocks], "timing": "median of three blocks of `steps` cycles",
            "mean_residual_contraction_per_cycle": (res1 / res0) ** (1.0 / (warmup + 3 * steps)) if res0 > 0 else 0.0,
            "residual_contraction_per_cycle_first_8": contraction8,
            "ms_per_residual_decade": ms_per_decade(dt * 1e3, contraction8),
            "device_memory_in_use_GB": (total_b - free_b) / 1e9,
            "device_memory_torch_tensors_GB": torch_gb,
            "release_setup_matrices": bool(release_setup_matrices),
            "device_memory_note": "in use = hipMemGetInfo after torch.cuda.empty_cache(): the library's buffers (inventory below), the "
                                  "tensors of the caller (mesh arrays of the LaplaceProblem, x, b), HIP / RCCL runtime",
            "device_memory_inventory_GB": {line[14:].strip(): float(line[:10]) for line in M.memory_inventory().splitlines() if line.strip()}}


def measure_cg_solve(ctx, torch, M, cells, params, reduction=1e-8, material="constant"):
    """The reference's driver workflow (tests/hierarchy_driver.cc:103-116): dealii::SolverCG on the fine operator with the
    hierarchy as preconditioner ("is preconditioner" true: every application starts from x = 0; the coarse cycle must be the
    symmetric V(1,1)).  Right-hand side: A times a random vector (a known solution), start x = 0, stopped at `reduction` of the
    initial residual (SolverControl takes the absolute norm).  Wall clock of the whole solve, one call through the C ABI."""
    prob = M.LaplaceProblem((cells,) * 3, material, device="cuda")
    p = json.loads(json.dumps(params))
    p["is preconditioner"] = True
    p["solver"].get("amg", {}).pop("pre_smoothing_levels", None)
    h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", prob, p)
    n = h.level_size(0)
    g = torch.Generator(device="cuda").manual_seed(3)
    free = (prob.constrained != 1).to(torch.float64)
    x_true = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * free
    b = torch.empty_like(x_true)
    h.operator_apply(0, x_true, b)
    b *= free
    r0 = ctx.l2_norm(b)
    out = None
    runs = []
    for attempt in range(4):            # (the first solve warms the launches up; the median of the three after it is reported)
        x = torch.zeros_like(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        its, hist = h.solve_cg(b, x, tolerance=reduction * r0, max_iterations=200)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if attempt > 0:
            runs.append(dt)
            dt = sorted(runs)[len(runs) // 2]
        err = float((x - x_true).abs().max() / x_true.abs().max())
        out = {"workload": f"{cells}^3 cells = {cells + 1}^3 DoFs, matrix-free, material {material}: CG preconditioned by one V-cycle "
                           f"(Chebyshev(3), symmetric V(1,1) coarse cycle), x0 = 0, stopped at ||r|| <= {reduction:g} ||r0||",
               "n_dofs": n, "iterations": int(its), "ms_total": dt * 1e3, "ms_per_iteration": dt * 1e3 / max(int(its), 1),
               "value": n / dt, "unit": "DoF/s solved to the tolerance", "residual_reduction": float(hist[-1] / hist[0]),
               "max_rel_error_vs_known_solution": err, "ms_total_runs": [v * 1e3 for v in runs]}
    return out


def measure_vcycle_f32(ctx, torch, M, h, prob, op_monitor_factory, steps=10, warmup=3):
    """BASELINE.json configs[4] (FP32) as a whole cycle: the fine level (smoother, residual) in FP32 through the FP32
    instance of the operator kernel, restriction / coarse solve / prolongation in FP64 (`Hierarchy.apply_f32`, the
    hierarchy of the headline measurement built with "fine level precision" float)."""
    n = h.level_size(0)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(n, dtype=torch.float32, device="cuda", generator=g)
    x *= (prob.constrained != 1).to(torch.float32)
    b = torch.zeros(n, dtype=torch.float32, device="cuda")
    op = op_monitor_factory()
    r = torch.empty(n, dtype=torch.float64, device="cuda")

    def norm():
        op.vmult(r, x.double())
        return ctx.l2_norm(r)
    r0 = norm()
    for _ in range(warmup):
        h.apply_f32(b, x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        h.apply_f32(b, x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # contraction over the first cycles only: a float iterate stalls at ~1e-7 of the start residual
    x = torch.rand(n, dtype=torch.float32, device="cuda", generator=g) * (prob.constrained != 1).to(torch.float32)
    r0 = norm()
    for _ in range(4):
        h.apply_f32(b, x)
    contraction = (norm() / r0) ** 0.25
    return {"workload": "the headline hierarchy with the fine level in FP32 (operator kernel, Chebyshev(3) smoother and "
                        "residual on float vectors; restriction, coarse solve and prolongation in FP64)",
            "n_dofs": n, "dtype": "f32 fine level / f64 coarse levels", "ms_per_step": dt * 1e3, "value": n / dt,
            "unit": "DoF/s", "mean_residual_contraction_per_cycle_first_4": contraction}


def measure_smoother_f32(ctx, torch, M, cells, degree, reps=5, material="linear"):
    """BASELINE.json configs[4] (FP32): smoother apply with the FP32 instance of the operator kernel (vector
    ALU: at ~11 flop/B the cell kernel sits below the FP32-MFMA ridge)."""
    prob = M.LaplaceProblem((cells,) * 3, material, device="cuda")
    op = M.MatrixFreeLaplaceF32(ctx, prob)
    compact = op.cell_constant_layout()
    N = prob.n_dofs
    del prob
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(N, dtype=torch.float32, device="cuda", generator=g)
    b = torch.zeros(N, dtype=torch.float32, device="cuda")
    s1, s2 = torch.empty_like(x), torch.empty_like(x)
    coefs = smoother_coefficients(degree, 0.09, 1.8)

    def apply():
        bufs = [s1, s2]
        cur, prev = x, None
        for k, (al, be) in enumerate(coefs):
            tgt = x if k == len(coefs) - 1 and len(coefs) > 1 else bufs[(len(coefs) - 2 - k) % 2]
            op.smoother_step(b, cur, prev, al, be, tgt)
            prev, cur = cur, tgt

    apply()
    ctx.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        apply()
    ev1.record()
    ev1.synchronize()
    ms = ev0.elapsed_time(ev1) / reps
    # bytes the layout requires in FP32: x 4 + out 4 + one id 4 + 8 (or 1) coefficients, + b + D^-1 (+ x_prev)
    per_dof = smoother_bytes_per_dof(len(coefs), 4, compact, survey=False, ids_computed=op.ids_computed())
    return {"n_dofs": N, "degree": degree, "dtype": "f32", "material": material,
            "coefficient_layout": "one value per cell" if compact else "eight values per cell",
            "ms_per_apply": ms, "required_bytes_per_dof": per_dof,
            "required_GBs": N * per_dof / (ms * 1e-3) / 1e9, "frac_of_8TBs": N * per_dof / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}


def measure_cell_contraction(ctx, torch, cells, reps=20):
    """BASELINE.json configs[4]: the cell-local evaluation as a batched dense contraction V = (K_ref U) diag(c) over
    cells^3 cells, planar operands, on the vector ALU and on the matrix cores, FP32 and FP64.  17 values move per cell
    (8 in, 1 coefficient, 8 out): HBM-bound either way; the MFMA issue time is beside it (two 16x16x4 MFMAs per 16 cells,
    half of each tile is padding: the operator is 8 x 8)."""
    n = cells ** 3
    out = {"n_cells": n, "bytes_per_cell": {"f32": 68, "f64": 136}}
    for dt, name in ((torch.float32, "f32"), (torch.float64, "f64")):
        u = torch.randn(8, n, dtype=dt, device="cuda")
        c = torch.rand(n, dtype=dt, device="cuda") + 1.0
        v = torch.empty_like(u)
        for variant in ("valu", "mfma"):
            for _ in range(3):
                ctx.cell_contraction(u, c, v, (1.0 / cells,) * 3, variant=variant)
            ctx.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
            ev[0].record()
            for r in range(reps):
                ctx.cell_contraction(u, c, v, (1.0 / cells,) * 3, variant=variant)
                ev[r + 1].record()
            ev[-1].synchronize()
            ts = sorted(ev[r].elapsed_time(ev[r + 1]) for r in range(reps))
            ms = ts[len(ts) // 2]
            gbs = n * 17 * u.element_size() / (ms * 1e-3) / 1e9
            out[f"{name}_{variant}"] = {"ms": ms, "GBs": gbs, "frac_of_8TBs": gbs / HBM_PEAK_GBS, "Gcells_per_s": n / (ms * 1e-3) / 1e9}
        del u, c, v
        torch.cuda.empty_cache()
    return out


def measure_smoother(ctx, torch, M, n_dofs_per_dim, degree, reps=20, warmup=3, tile=None, material="constant",
                     stored_diagonal=False):
    """Fine-level smoother apply (degree fused operator kernels) on its own: one HIP event pair per apply on the
    kernels' stream, `warmup` untimed applies, then `reps` timed ones; reported: median (the quoted figure), min,
    max.  Rates: `required_GBs` on the bytes the layout makes the kernel read at least (see
    required_bytes_per_dof), `survey_GBs` on the SURVEY.md 8(d) figure (8 index ints and 8 coefficients per DoF)."""
    prob = M.LaplaceProblem((n_dofs_per_dim - 1,) * 3, material, device="cuda")
    ctx.set_stored_diagonal(stored_diagonal)
    op = M.MatrixFreeLaplace(ctx, prob)
    ctx.set_stored_diagonal(False)
    compact = op.cell_constant_layout()
    dinv_stored = op.diagonal_in_record()
    if tile:
        op.set_tile(*tile)
    N = prob.n_dofs
    del prob
    torch.cuda.empty_cache()
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(N, dtype=torch.float64, device="cuda", generator=g)
    b = torch.zeros(N, dtype=torch.float64, device="cuda")
    s1, s2 = torch.empty_like(x), torch.empty_like(x)
    coefs = smoother_coefficients(degree, 0.09, 1.8)

    def apply_terms():
        # x_{k+1} targets alternate so that the last term lands in x (HipSmoother::apply, one launch per term)
        bufs = [s1, s2]
        cur, prev = x, None
        for k, (al, be) in enumerate(coefs):
            tgt = x if k == len(coefs) - 1 and len(coefs) > 1 else bufs[(len(coefs) - 2 - k) % 2]
            op.smoother_step(b, cur, prev, al, be, tgt)
            prev, cur = cur, tgt

    state = {"cur": x, "other": s1}

    def apply_sweep():
        # the whole polynomial in one sweep, out of place (HipSmoother::apply_to: the hierarchy alternates two vectors)
        op.smoother_sweep([c[0] for c in coefs], [c[1] for c in coefs], b, state["cur"], state["other"])
        state["cur"], state["other"] = state["other"], state["cur"]

    def timed(apply):
        for _ in range(warmup):
            apply()
        ctx.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record()
        for r in range(reps):
            apply()
            ev[r + 1].record()
        ev[-1].synchronize()
        return sorted(ev[r].elapsed_time(ev[r + 1]) for r in range(reps))

    sweep = 2 <= len(coefs) <= 3 and op.sweep_available(len(coefs))
    ts_terms = timed(apply_terms)
    ts = timed(apply_sweep) if sweep else ts_terms
    ms = ts[len(ts) // 2]
    survey = survey_8d_bytes_per_dof(len(coefs))["smoother_apply"]
    required = smoother_bytes_per_dof(len(coefs), 8, compact, survey=False, dinv_stored=dinv_stored, ids_computed=op.ids_computed())
    fused_form = 8 * (4 + (1 if dinv_stored else 0)) if sweep else None     # x_0, b, one coefficient per cell (D^-1), x_K
    traffic = None
    if dinv_stored == (not compact):
        traffic = (committed_traffic(n_dofs_per_dim, degree, compact, op.get_sweep_tile(len(coefs)), prefix="sweep_dofs") if sweep else
                   committed_traffic(n_dofs_per_dim, degree, compact, op.get_tile(), prefix="dofs"))
    return {"n_dofs": N, "degree": degree, "material": material,
            "coefficient_layout": "one value per cell" if compact else "eight values per cell",
            "diagonal": "D^-1 stored in the chunk records" if dinv_stored else "D^-1 derived in the kernel from the cell coefficients (not read)",
            "kernel": (f"mf_cheb_fused_kernel: the {len(coefs)} terms in one sweep" if sweep else f"mf_laplace_kernel: {len(coefs)} launches, one per term"),
            "tile_waves_ty_tz": list(op.get_sweep_tile(len(coefs)) if sweep else op.get_tile()),
            "hbm_traffic_bytes_per_launch_pmc": traffic,
            "hbm_traffic_GBs": (traffic * (1 if sweep else len(coefs)) / (ms * 1e-3) / 1e9) if traffic else None,
            "ms_per_apply": ms, "ms_min": ts[0], "ms_max": ts[-1], "reps": reps, "warmup": warmup,
            "ms_per_apply_term_by_term": ts_terms[len(ts_terms) // 2],
            "required_bytes_per_dof": required,
            "required_bytes_are": "the sum over the polynomial terms of what the layout requires of a launch of that term (x, out, one "
                                  "id, coefficients, b, x_prev, D^-1 where stored) -- the figure of rounds 1-3, whichever kernel runs",
            "required_GBs": N * required / (ms * 1e-3) / 1e9,
            "frac_of_8TBs": N * required / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "fused_form_bytes_per_dof": fused_form,
            "fused_form_GBs": (N * fused_form / (ms * 1e-3) / 1e9) if fused_form else None,
            "survey_8d_bytes_per_dof_smoother_apply": survey, "survey_8d_GBs": N * survey / (ms * 1e-3) / 1e9}


def cpu_baseline(args, M, h, prob, lmin, lmax, torch):
    """The oracle's C++/OpenMP restatement ("port") of the same V-cycle on the host cores -- and, since it runs the
    very workload of the headline, the parity check at that size: the iterates the timed CPU cycles produce are kept
    and compared with the GPU cycle started from the same vector (`parity_vs_gpu`)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle_native as ON

    ctx = h.ctx
    n = prob.n
    cd = prob.cell_dofs.cpu().numpy()
    co = prob.coefficient.cpu().numpy()
    cn = prob.constrained.cpu().numpy()
    R = h.restrictor().to_scipy()
    Ac = h.coarse_operator().to_scipy()
    cores = ON.effective_cpu_count()
    ON.set_num_threads(cores)
    # D^-1 of the matrix-free operator (constrained entries one): the oracle's own compute_diagonal
    dinv = 1.0 / ON.mf_diagonal(n, prob.h, cd, co, cn)
    rng = np.random.default_rng(0)
    x0 = np.where(cn.astype(bool), 0.0, rng.random(prob.n_dofs))
    b = np.zeros(prob.n_dofs)
    amg = h.coarse_amg_levels() if args.coarse == "amg" else None
    pre_levels = args.amg_pre_levels if args.amg_pre_levels >= 0 else None
    def run(x_in, cycles):
        xx, _ = ON.vcycles(n, prob.h, cd, co, cn, dinv, args.degree, lmin, lmax, R, Ac, args.coarse_iters, b, x_in, cycles,
                           want_history=False, amg_levels=amg, amg_pre_smoothing_levels=pre_levels)
        return xx

    # all host cores, 2 warm-ups, >= 5 timed cycles, median; plus a 1-thread figure (the
    # reference's own tests run one thread per MPI rank)
    x = run(x0, 2)
    iterates = {2: x}
    times = []
    for c in range(args.cpu_cycles):
        t0 = time.perf_counter()
        x = run(x, 1)
        times.append(time.perf_counter() - t0)
        iterates[3 + c] = x
    dt = sorted(times)[len(times) // 2]
    ON.set_num_threads(1)
    t0 = time.perf_counter()
    run(x, 1)
    dt1 = time.perf_counter() - t0
    ON.set_num_threads(cores)
    del R, Ac, amg
    # parity at this size (harness: /root/reference/tests/test_hierarchy.cc:95-123, b = 0): the residual norm
    # ||A x_k|| / ||A x_0|| of every kept CPU iterate (the oracle's operator) against the GPU cycle's (the product's
    # operator), and the iterates themselves
    resn = lambda v: float(np.linalg.norm(ON.mf_apply(n, prob.h, cd, co, cn, v)))
    r0_cpu = resn(x0)
    op = M.MatrixFreeLaplace(ctx, prob)
    xg = torch.from_numpy(x0).cuda()
    bg = torch.zeros_like(xg)
    rg = torch.empty_like(xg)
    op.vmult(rg, xg)
    r0_gpu = ctx.l2_norm(rg)
    hist_diff = iter_diff = 0.0
    hist_cpu, hist_gpu = [], []
    for k in range(1, max(iterates) + 1):
        h.apply(bg, xg)
        if k in iterates:
            op.vmult(rg, xg)
            hg, hc = ctx.l2_norm(rg) / r0_gpu, resn(iterates[k]) / r0_cpu
            hist_cpu.append(hc)
            hist_gpu.append(hg)
            hist_diff = max(hist_diff, abs(hg - hc) / hc)
            xc = torch.from_numpy(iterates[k]).cuda()
            iter_diff = max(iter_diff, float((xg - xc).abs().max().item()))
            del xc
    del op, xg, bg, rg
    return {"value": prob.n_dofs / dt, "unit": "DoF/s", "cores": cores, "kind": "port",
            "sample": f"median of {args.cpu_cycles} V-cycles (after 2 warm-ups) of the same {prob.N[0]}^3-DoF workload, "
                      f"oracle/oracle_kernels.cpp with OpenMP on {cores} host threads",
            "ms_per_step": dt * 1e3,
            "single_thread": {"value": prob.n_dofs / dt1, "unit": "DoF/s", "cores": 1, "ms_per_step": dt1 * 1e3,
                              "sample": "1 V-cycle of the same workload on one thread"},
            "parity_vs_gpu": {"what": "the iterates of the CPU cycles above (same start vector, b = 0) against the GPU "
                                      "cycle: relative residual history ||A x_k|| / ||A x_0|| and the iterates, cycles "
                                      f"{min(iterates)}..{max(iterates)}",
                              "max_rel_history_diff_vs_gpu": hist_diff,
                              "max_abs_iterate_diff_vs_gpu": iter_diff,
                              "history_cpu": hist_cpu, "history_gpu": hist_gpu}}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and os.environ.get("OMP_NUM_THREADS") in (None, "", "1"):
        # torch.distributed.run gives every worker ONE OpenMP thread; the host parts of the setup (sorting, class detection, CSR
        # assembly) are parallel: a share of the node's cores per rank, never more than there are (spinning teams of eight ranks on
        # too few cores starve the threads that launch kernels: scratch/rank_cycle_on_one_gpu.py), and idle teams sleep
        os.environ["OMP_NUM_THREADS"] = str(max(1, min(16, (os.cpu_count() or world) // world)))
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU")
    backend = os.environ.get("MFMG_BENCH_BACKEND", "nccl")   # "gloo": rehearsal with ranks sharing a card
    device_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    import mfmg_amd as M

    ctx = M.Context()
    n = args.cells + 1
    t_setup = time.perf_counter()
    # weak scaling: every GPU owns cells^3 cells; the global box doubles in z, y, x in turn, so that N = 8 is
    # the cube with 2*cells per direction (BASELINE.json configs[3]); slabs are cut along z
    gx, gy, gz = args.cells, args.cells, args.cells
    for i in range(int(round(math.log2(world)))):
        if i % 3 == 0:
            gz *= 2
        elif i % 3 == 1:
            gy *= 2
        else:
            gx *= 2
    if args.box:
        gx, gy, gz = (int(v) for v in args.box.split(","))
    part = transport = None
    if world > 1:
        if world & (world - 1):
            raise SystemExit("--gpus must be a power of two")
        # boxes: the ranks sit where the weak-scaling mesh grew (z, then y, then x), one box of cells^3 each
        rank_grid = (1, 1, world)
        if args.grid:
            rank_grid = tuple(int(v) for v in args.grid.split(","))
        elif args.partition == "box":
            rank_grid = [1, 1, 1]
            for i in range(int(round(math.log2(world)))):
                rank_grid[2 - i % 3] *= 2
            rank_grid = tuple(rank_grid)
        assert rank_grid[0] * rank_grid[1] * rank_grid[2] == world, "--grid must multiply to --gpus"
        part = M.BoxPartition((gx, gy, gz), rank, rank_grid, length=(gx / args.cells, gy / args.cells, gz / args.cells),
                              low_ghost_cells=args.low_ghost)
        if backend != "nccl":
            os.environ.setdefault("MFMG_BENCH_TRANSPORT", "host")
        transport = M.HaloTransport(ctx, part, 2, transport=os.environ.get("MFMG_BENCH_TRANSPORT") or None)
        prob = part.local_problem(args.material, device="cuda")
    elif args.box:
        # one rank on the global box of a distributed run (rehearsals: the distributed cycle must contract identically)
        prob = M.LaplaceProblem((gx, gy, gz), args.material, device="cuda",
                                cell_size=(1.0 / args.cells,) * 3)
    else:
        prob = M.LaplaceProblem((args.cells,) * 3, args.material, device="cuda")
    params = {
        "eigensolver": {"number of eigenvectors": 2},
        "agglomeration": {"partitioner": "block", "nx": 2, "ny": 2, "nz": 2},
        "smoother": {"type": "Chebyshev", "degree": args.degree, "smoothing_range": 20.0, "n_smoothing_steps": 1},
        "solver": ({"type": "pcg", "n_iterations": args.coarse_iters} if args.coarse == "pcg" else
                   {"type": "amg", "amg": {"smoother_degree": args.amg_degree, "smoothing_range": 4.0, "n_cycles": 1,
                                           "aggregate_block": args.amg_block,
                                           **({"setup": args.amg_setup} if args.amg_setup else {}),
                                           **({"pre_smoothing_levels": args.amg_pre_levels} if args.amg_pre_levels >= 0 else {}),
                                           **({"replicate_rows": args.amg_replicate_rows} if args.amg_replicate_rows else {}),
                                           **({"deep_level": int(args.amg_deep.split(",")[0]),
                                               "deep_block": int(args.amg_deep.split(",")[1])} if args.amg_deep else {})}}),
        "is preconditioner": False,
        "max levels": 2,
    }
    evaluator = "HipMatrixFreeMeshEvaluator" if args.evaluator == "matrix_free" else "HipMeshEvaluator"
    # (the FP32 instance of the fine operator rides along for the extra line `vcycle_fp32_fine_level_config5`;
    # the headline cycle below is the FP64 `apply`)
    with_f32 = world == 1 and args.evaluator == "matrix_free" and not args.no_extras
    torch.cuda.synchronize()
    t_problem = time.perf_counter() - t_setup          # mesh arrays of the synthetic problem (torch), partition, transport
    t_setup = time.perf_counter()
    h = M.Hierarchy(ctx, evaluator, prob, dict(params, **{"fine level precision": "float"}) if with_f32 else params)
    ctx.synchronize()
    t_setup = time.perf_counter() - t_setup            # the Hierarchy constructor: what the reference's "Setup" timer covers
    degree, lmin, lmax = h.smoother_info()
    n_local, n_coarse = h.level_size(0), h.level_size(1)
    # DoFs this rank owns (the local vector also holds the ghost planes of the neighbours)
    n_fine = n_local if part is None else part.own_n[0] * part.own_n[1] * part.own_n[2]
    n_global = n_fine if part is None else part.n_global_dofs

    # the same global start vector whatever the number of ranks (a rank cuts its slab out of it)
    g = torch.Generator(device="cuda").manual_seed(1)
    if part is None:
        x = torch.rand(n_local, dtype=torch.float64, device="cuda", generator=g)
    else:
        x = part.local_from_global(torch.rand(part.n_global_dofs, dtype=torch.float64, device="cuda", generator=g))
        torch.cuda.empty_cache()
    x *= (prob.constrained != 1).to(torch.float64)
    b = torch.zeros(n_local, dtype=torch.float64, device="cuda")

    def barrier():
        if world > 1:
            dist.barrier()

    # residual norms around the run: a bench of a cycle that does not contract would be meaningless
    assembled = args.evaluator == "assembled"
    op_monitor = None if assembled else M.MatrixFreeLaplace(ctx, prob)
    compact = (not assembled) and op_monitor.cell_constant_layout()   # same detection as inside the hierarchy
    ids_computed = (not assembled) and op_monitor.ids_computed()
    r = torch.empty_like(x)

    def residual_norm():
        if assembled and transport is None:
            h.operator_apply(0, x, r)      # b = 0
            return ctx.l2_norm(r)
        if transport is None:
            op_monitor.vmult(r, x)
            ctx.sadd(r, -1.0, 1.0, b)
            return ctx.l2_norm(r)
        h.operator_apply(0, x, r)          # with halo exchange; b = 0
        return transport.owned_norm(r)

    res_start = residual_norm()
    if args.tile and not assembled:
        _t = [int(v) for v in args.tile.split(',')]
        h.set_operator_tile(_t[2] if len(_t) > 2 else 0, _t[0], _t[1])
    mf_tile = (0, 0, 0) if assembled else h.operator_tile()
    sweep_terms = (0, 0) if assembled else h.smoother_sweep_terms()
    sweep_tile = h.sweep_tile(sweep_terms[1]) if sweep_terms[1] else None
    for _ in range(args.warmup):
        h.apply(b, x)
    # HIP events around the launches of the dominant kernel only: an event pair costs ~5 us on the stream, and the
    # cycle has ~45 launches; the other kernel family is timed in a separate pass after the timed region
    dominant = "csr_spmv_kernel" if assembled else "mf_laplace_kernel,mf_cheb_fused_kernel"
    ctx.profile_enable(True, only=dominant)
    n_ex0 = transport.n_exchanges() if transport is not None else 0
    n_vol0 = transport.exchange_volume() if transport is not None else 0
    n_ov0 = transport.n_overlapped() if transport is not None else 0
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        h.apply(b, x)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    n_exchanges_per_cycle = ((transport.n_exchanges() - n_ex0) / max(args.steps, 1)) if transport is not None else 0.0
    mb_sent_per_cycle = ((transport.exchange_volume() - n_vol0) * 8e-6 / max(args.steps, 1)) if transport is not None else 0.0
    n_overlapped_per_cycle = ((transport.n_overlapped() - n_ov0) / max(args.steps, 1)) if transport is not None else 0.0
    launches, k_ms, k_bytes = ctx.profile_query("mf_laplace_kernel")
    f_launches, f_ms, f_bytes = ctx.profile_query("mf_cheb_fused_kernel")
    fused_dominant = f_ms > k_ms
    one_term = {"launches": launches, "total_ms": k_ms, "algorithmic_bytes": k_bytes}
    if fused_dominant:
        launches, k_ms, k_bytes = f_launches, f_ms, f_bytes
    c_launches, c_ms, c_bytes = ctx.profile_query("csr_spmv_kernel")
    ctx.profile_enable(False)      # (the launches of the timed region only: the passes below are not part of it)
    res_end = residual_norm()
    contraction = (res_end / res_start) ** (1.0 / max(args.warmup + args.steps, 1)) if res_start > 0 else 0.0
    # ... and over the first 8 cycles from a fresh start (the timed run iterates on: after ~25 cycles its residual sits on the
    # rounding floor and the mean over all of them understates the contraction per cycle)
    x_keep = x.clone()
    g8 = torch.Generator(device="cuda").manual_seed(2)
    if part is None:
        x.copy_(torch.rand(n_local, dtype=torch.float64, device="cuda", generator=g8))
    else:
        x.copy_(part.local_from_global(torch.rand(part.n_global_dofs, dtype=torch.float64, device="cuda", generator=g8)))
    x *= (prob.constrained != 1).to(torch.float64)
    r8_0 = residual_norm()
    for _ in range(8):
        h.apply(b, x)
    r8_1 = residual_norm()
    contraction8 = (r8_1 / r8_0) ** 0.125 if r8_0 > 0 else 0.0
    x.copy_(x_keep)
    del x_keep
    del op_monitor, r
    other_cycles = 0
    if not assembled:
        # second pass, outside the timed region: the SpMV family (x and b keep converging; timings do not depend on it)
        other_cycles = 3
        ctx.profile_enable(True, only="csr_spmv_kernel")
        for _ in range(other_cycles):
            h.apply(b, x)
        torch.cuda.synchronize()
        c_launches, c_ms, c_bytes = ctx.profile_query("csr_spmv_kernel")
        ctx.profile_enable(False)
    # third pass: the one-pass residual restriction b_c = R (A x - b), where the hierarchy uses it
    rr_launches = rr_ms = rr_bytes = 0
    rr_classes = h.residual_restriction_classes() if h.n_levels > 1 else 0
    if rr_classes > 0:
        ctx.profile_enable(True, only="residual_restriction")
        for _ in range(3):
            h.apply(b, x)
        torch.cuda.synchronize()
        rr_launches, rr_ms, rr_bytes = ctx.profile_query("residual_restriction")
        ctx.profile_enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = n_global / (dt / args.steps)

    coarse_desc = (f"{args.coarse_iters} Jacobi-PCG steps" if args.coarse == "pcg" else
                   "one V-cycle of a smoothed-aggregation hierarchy (Chebyshev(1) = damped-Jacobi smoothers, dense LU at the bottom; "
                   + ("V(1,1) on every level" if args.amg_pre_levels < 0 else
                      f"V(0,1) -- post-smoothing only -- from level {args.amg_pre_levels} of that hierarchy on") + ")")
    if rank == 0:
        # bytes the layout requires per launch (library accounting: x, out, one id, coefficients + epilogue operands)
        achieved = (k_bytes / launches) / (k_ms / launches * 1e-3) / 1e9 if launches else 0.0
        word = 8
        out = {
            "metric": "fine-DoFs/sec per V-cycle (3D Laplace)",
            "value": value,
            "unit": "DoF/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"3D Laplace on the unit cube, Q1 {'assembled CSR (BASELINE.json configs[2])' if assembled else 'matrix-free'}, {args.cells}^3 cells = {n}^3 DoFs per GPU "
                            f"(the mesh deal.II's refine_global gives; BASELINE.json configs[3] '512^3 on 8 GPUs' "
                            f"is this workload at N=8), spectral AMGe (2x2x2 agglomerates, 2 eigenvectors), "
                            f"Chebyshev({degree}) smoother, coarse level {n_coarse} DoFs: " + coarse_desc + ", FP64",
                "material": args.material,
                "coefficient_layout": (None if assembled else
                                       ("one value per cell (the eight quadrature coefficients of every cell are equal)"
                                        if compact else "eight values per cell")),
                "fine_dofs_per_gpu": n_fine,
                "coarse_dofs_per_gpu": n_coarse,
                "coarse_amg_levels_rows_nnzA_nnzP": (h.coarse_amg_shapes() if args.coarse == "amg" else None),
                "smoother": {"type": "Chebyshev", "degree": degree, "lambda_min": lmin, "lambda_max": lmax},
                "parallelism": "1 GPU" if world == 1 else
                               f"{world} GPUs, {'x'.join(map(str, part.grid))} ranks (x, y, z) on a {gx}x{gy}x{gz}-cell box "
                               f"({'boxes' if part.split_xy else 'slabs along z'}), halo exchange "
                               f"per operator application on every level of the cycle (transport: {transport.name()}, "
                               f"{n_exchanges_per_cycle:.1f} exchanges per cycle and rank, each ONE grouped send/recv with all "
                               f"neighbours, {n_overlapped_per_cycle:.1f} of them beside operator tiles on a second stream, "
                               f"{mb_sent_per_cycle:.2f} MB sent per cycle by rank 0), aggregation levels coupled "
                               f"across the ranks, the levels from {h.coarse_amg_gather_rows()} global rows down gathered and solved "
                               f"redundantly: the same preconditioner as on one GPU",
                "global_dofs": n_global,
                "setup_seconds": t_setup,
                "setup_seconds_covers": "the Hierarchy constructor (operator layouts, smoother bounds, restrictor, R A R^T, coarse solver"
                                        + (", the FP32 instance of the fine operator" if with_f32 else "") + ")",
                "problem_seconds": t_problem,
                "mean_residual_contraction_per_cycle": contraction,
                "residual_contraction_per_cycle_first_8": contraction8,
                "ms_per_residual_decade": ms_per_decade(ms_per_step, contraction8),
                "coarse_cycle": ("V(1,1) on every level of the aggregation hierarchy (the library default, symmetric: what a CG-"
                                 "preconditioned use needs)" if args.amg_pre_levels < 0 or args.coarse != "amg" else
                                 f"V(0,1) from level {args.amg_pre_levels} of the aggregation hierarchy on (post-smoothing only: not a "
                                 "symmetric preconditioner; the symmetric V(1,1) cycle is measured beside it: "
                                 "vcycle_256cubed_coarse_cycle_v11, with ms_per_residual_decade for both)"),
                "rccl_ranks": (transport.comm_ranks() if transport is not None else None),
            },
            "roofline": {
                "kernel": (f"mf_cheb_fused_kernel (the {degree} terms of the Chebyshev smoother over the matrix-free operator in ONE sweep)"
                           if fused_dominant else "mf_laplace_kernel (fused matrix-free operator + smoother/residual epilogue)"),
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": (None if assembled else
                            committed_traffic(args.cells, args.degree, compact, sweep_tile, prefix="sweep_cells") if fused_dominant else
                            committed_traffic(args.cells, args.degree, compact, mf_tile)),
                "traffic_source": (None if assembled else
                                   committed_traffic(args.cells, args.degree, compact, sweep_tile, prefix="sweep_cells", want_source=True)[1]
                                   if fused_dominant else committed_traffic(args.cells, args.degree, compact, mf_tile, want_source=True)[1]),
                "priced_on": "algorithmic bytes of the smoother terms a launch performs: per term what the data layout requires of a "
                             "launch of that term (x, out, one id, coefficients, b, x_prev, and D^-1 where the layout stores it: eight "
                             "coefficients per cell); halo re-reads of the tiling are waste and not counted"
                             + ("; the sweep performs all terms per launch, the bytes its fused form must move are `fused_form_*`" if fused_dominant else ""),
                "terms_per_launch": (degree if fused_dominant else 1),
                "fused_form_bytes_per_launch": (32.0 * n_local if fused_dominant else None),
                "fused_form_GBs": ((32.0 * n_local) / (k_ms / launches * 1e-3) / 1e9 if fused_dominant and launches else None),
                "one_term_kernel_in_timed_region": (one_term if fused_dominant else None),
                "required_bytes_per_dof_operator": (None if assembled else operator_bytes_per_dof(word, compact, False, ids_computed)),
                "ids": (None if assembled else ("computed by the kernel (structured numbering)" if ids_computed
                                                else "one 4-byte id per DoF read from the chunk records")),
                "survey_8d_bytes_per_dof": (None if assembled else survey_8d_bytes_per_dof(degree)),
                "survey_8d_GBs": (None if assembled or not launches else      # SURVEY.md 8(d)'s own figure for what one launch does
                                  survey_8d_bytes_per_dof(degree)["smoother_apply" if fused_dominant else "smoother_term"] * n_local
                                  / (k_ms / launches * 1e-3) / 1e9),
                "launches_in_timed_region": launches, "tile_waves_ty_tz": list(sweep_tile if fused_dominant else mf_tile),
                "avg_launch_ms": k_ms / launches if launches else None,
                "required_bytes_per_launch": k_bytes / launches if launches else None,
                "share_of_step_time": k_ms / (ms_per_step * args.steps) if launches else None,
            },
            "other_kernels": {
                "csr_spmv_kernel": {"what": "coarse-level family: A_c and the levels below, R, R^T, prolongators (table-driven "
                                            "layouts read no matrix values, so no byte rate is quoted; HBM bytes per launch "
                                            "from the PMC passes: profiles/r03_g_cycle_hbm_bytes_per_launch.txt)",
                                    "launches": c_launches, "total_ms": c_ms,
                                    "share_of_step_time": (c_ms / (ms_per_step * (other_cycles or args.steps))) if c_ms else None,
                                    "timed_in": (f"{other_cycles} extra cycles after the timed region" if other_cycles
                                                 else "the timed region")},
                "residual_restriction_kernel": (None if not rr_launches else {
                    "what": "b_c = R (A x - b) in one pass over x and b (the residual of the cycle is never stored): the rows of "
                            "R A repeat themselves from agglomerate to agglomerate and come from tables",
                    "agglomerate_classes": rr_classes, "launches": rr_launches, "avg_launch_ms": rr_ms / rr_launches,
                    "algorithmic_bytes_per_launch": rr_bytes / rr_launches,
                    "GBs_on_x_b_and_b_c": (rr_bytes / rr_launches) / (rr_ms / rr_launches * 1e-3) / 1e9,
                    "timed_in": "3 extra cycles after the timed region"}),
            },
        }
        tile = tuple(int(v) for v in args.tile.split(",")) if args.tile else None
        if assembled:
            # the dominant kernel of the assembled path is the SpMV family, priced at SURVEY.md 8(d)'s CSR figure
            rf = out["roofline"]
            ach = (c_bytes / (c_ms * 1e-3) / 1e9) if c_ms else 0.0
            rf.update({"kernel": "csr_spmv_kernel family (fine operator with fused smoother epilogues, R, R^T, coarse levels)",
                       "launches_in_timed_region": c_launches,
                       "avg_launch_ms": c_ms / c_launches if c_launches else None,
                       "share_of_step_time": c_ms / (ms_per_step * args.steps) if c_ms else None,
                       "priced_on": "SURVEY.md 8(d) CSR figure: 12 B per entry + 4 B per row + vectors"})
            if ach > HBM_PEAK_GBS:
                rf.update({"achieved": None, "frac": None,
                           "note": "with a constant coefficient the rows of the fine matrix and of the coarse operators "
                                   "repeat a few stencils and are evaluated from tables (no matrix values or column "
                                   "indices are read): the CSR-priced rate would exceed the HBM peak and is not a "
                                   "fraction of it -- see --material linear for stored values"})
            else:
                rf.update({"achieved": ach, "frac": ach / HBM_PEAK_GBS})
        if world == 1 and not args.no_cpu_baseline and not assembled:
            out["cpu_baseline"] = cpu_baseline(args, M, h, prob, lmin, lmax, torch)
        if world == 1 and not args.no_extras:
            try:
                # (what the partition of a distributed run costs a rank is measured by scratch/rank_cycle_on_one_gpu.py -- eight ranks as
                # threads, one of them timed alone behind the reflecting transport: DESIGN.md 7, profiles/r04_h -- not here: the leg of
                # round 3 that emulated the launch structure of a term-by-term smoother no longer describes a rank, whose whole
                # Chebyshev(3) smoother is one sweep)
                if with_f32:
                    out["vcycle_fp32_fine_level_config5"] = measure_vcycle_f32(ctx, torch, M, h, prob,
                                                                              lambda: M.MatrixFreeLaplace(ctx, prob))
                h = x = b = None
                torch.cuda.empty_cache()
                # the same cycle and smoother without the redundancies of the constant material (eight coefficients per
                # cell, every coarse-operator row and restrictor block stored): what a variable coefficient gets
                gen = measure_vcycle_small(ctx, torch, M, args.cells, params, material="linear")
                gen["smoother_apply"] = measure_smoother(ctx, torch, M, args.cells + 1, args.degree, material="linear")
                # the same with the setup's matrices rounded to float ("setup value precision" float: the stored blocks of R,
                # R A R^T and the aggregation hierarchy take half the bytes; arithmetic FP64; the cycle is the exact cycle of
                # the rounded matrices -- tests/test_gpu_hierarchy.py::test_setup_value_precision_float)
                gen["setup_value_precision_float"] = measure_vcycle_small(
                    ctx, torch, M, args.cells, dict(params, **{"setup value precision": "float"}), material="linear")
                out["general_coefficient"] = gen
                out["vcycle_128cubed_config1"] = measure_vcycle_small(ctx, torch, M, 128, params)
                if args.coarse == "amg" and args.amg_pre_levels >= 0:
                    # the headline workload with the symmetric coarse cycle of rounds 1 and 2 (V(1,1) on every level of the
                    # aggregation hierarchy: what a CG-preconditioned use needs), for continuity
                    p11 = json.loads(json.dumps(params))
                    p11["solver"]["amg"].pop("pre_smoothing_levels", None)
                    out["vcycle_256cubed_coarse_cycle_v11"] = measure_vcycle_small(ctx, torch, M, args.cells, p11)
                    # ... and what the reference's driver does with it: a CG solve preconditioned by that cycle
                    try:
                        out["cg_solve_256cubed"] = measure_cg_solve(ctx, torch, M, args.cells, params)
                    except Exception as e:  # noqa: BLE001 - an extra leg: reported, never fatal for the headline
                        out["cg_solve_256cubed"] = {"error": f"{type(e).__name__}: {e}"[:300]}
                out["smoother_apply_256cubed_f32_config5"] = measure_smoother_f32(ctx, torch, M, 256, args.degree)
                out["cell_contraction_256cubed_config5"] = measure_cell_contraction(ctx, torch, 256)
                if not assembled:
                    out["vcycle_256cubed_assembled_config2"] = measure_vcycle_small(
                        ctx, torch, M, args.cells, params, evaluator="HipMeshEvaluator")
            except Exception as e:  # noqa: BLE001
                if not args.allow_missing_extras:
                    raise
                out["extras_error"] = str(e)
        # ---- the GLOBAL problem of BASELINE.json configs[3] (512^3 cells = 513^3 DoFs) on ONE GPU: the strong-scaling anchor
        #      of the 8-GPU run and the "512^3 V-cycle" figure; fits the 288 GB of one MI355X
        if world == 1 and not args.no_vcycle_513 and not args.no_extras and not assembled:
            h = x = b = None
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            try:
                out["vcycle_513cubed_1gpu"] = measure_vcycle_small(ctx, torch, M, 512, params, steps=5, warmup=2, release_setup_matrices=True)
            except Exception as e:  # noqa: BLE001
                if not args.allow_missing_extras:
                    raise
                out["vcycle_513cubed_1gpu"] = {"error": str(e)}
        # ---- north_star target legs: the fine-level smoother apply at 512^3 DoFs, measured with nothing else resident (the
        #      hierarchies of the other legs are freed first: with ~25 GB of them still allocated the same launches ran 4 %
        #      slower).  The default cell-constant layout derives D^-1 in the kernel: fewer bytes, faster, and a lower byte
        #      RATE; the variant that keeps D^-1 in the records is measured beside it.
        if world == 1 and not args.no_smoother_512:
            h = x = b = None
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            for mat, key, stored in (("constant", "smoother_apply_512cubed", False),
                                     ("constant", "smoother_apply_512cubed_stored_diagonal", True),
                                     ("linear", "smoother_apply_512cubed_general_coefficient", False)):
                try:
                    out[key] = measure_smoother(ctx, torch, M, 512, args.degree, tile=tile, material=mat, stored_diagonal=stored)
                except Exception as e:  # noqa: BLE001
                    if not args.allow_missing_extras:
                        raise
                    out[key] = {"error": str(e)}
            legs = [out[k] for k in ("smoother_apply_512cubed", "smoother_apply_512cubed_stored_diagonal",
                                     "smoother_apply_512cubed_general_coefficient") if "frac_of_8TBs" in out[k]]
            out["north_star_512cubed_smoother"] = {
                "target": ">= 0.50 of the 8 TB/s HBM3E peak on the fine-level smoother apply, 512^3 DoFs, 1 GPU; priced on the bytes "
                          "each layout requires",
                "frac_by_layout": {leg["material"] + " / " + leg["diagonal"]: leg["frac_of_8TBs"] for leg in legs},
                "ms_by_layout": {leg["material"] + " / " + leg["diagonal"]: leg["ms_per_apply"] for leg in legs}}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
