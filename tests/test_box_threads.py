"""The 2 x 2 x 2 box grid of SURVEY.md 8(e) through the LIBRARY path on one GPU: eight ranks as eight threads of one process
(the pool allows six processes on a card, so the multi-process tests of tests/test_distributed.py stop at four ranks), each
with its own context and hierarchy; the host transport's callbacks hand the messages over through in-memory mailboxes.
Every rank has seven neighbours -- three faces, three edges, one corner -- in every exchange.  The 20-cycle residual
history and the final iterate must equal the single-process hierarchy on the global mesh, and the oracle's restatement."""
import os
import queue
import sys
import threading

import numpy as np
import pytest
import torch

import mfmg_amd as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mfmg_oracle as O  # noqa: E402


class Mailboxes:
    """Transport between the threads of one process: a queue per ordered pair of ranks, a barrier for the collectives."""

    def __init__(self, n):
        self.n = n
        self.q = {(a, b): queue.Queue() for a in range(n) for b in range(n)}
        self.barrier = threading.Barrier(n, timeout=300)
        self.slots = [None] * n

    def callbacks(self, rank):
        def exchange(peers, send, recv):
            for p, s in zip(peers, send):
                self.q[(rank, p)].put(np.array(s, copy=True))
            for p, r in zip(peers, recv):
                r[:] = self.q[(p, rank)].get(timeout=300)

        def allreduce(values, op):
            self.slots[rank] = np.array(values, copy=True)
            self.barrier.wait()
            res = np.max(self.slots, axis=0) if op == 1 else np.sum(self.slots, axis=0)
            self.barrier.wait()          # nobody overwrites a slot before everybody has read it
            values[:] = res

        def allgather(src, out):
            self.slots[rank] = np.array(src, copy=True)
            self.barrier.wait()
            out[:] = np.concatenate(self.slots)
            self.barrier.wait()
        return exchange, allreduce, allgather


@pytest.mark.gpu
@pytest.mark.parametrize("material,amg,low_ghost", [("linear", {"coarsest_size": 40, "replicate_rows": 40, "pre_smoothing_levels": 0}, 2),
                                                    ("constant", {"coarsest_size": 300}, 2),
                                                    ("constant", {"coarsest_size": 40, "replicate_rows": 40, "pre_smoothing_levels": 0}, 4)])
def test_box_2x2x2_eight_ranks_in_one_process(mfmg_lib, material, amg, low_ghost):
    """low_ghost = 4: two agglomerates of every lower neighbour in the local mesh -- the Chebyshev(3) smoother of every rank is
    one sweep with one exchange of x, three ghost planes deep (asserted below)."""
    grid, per = (2, 2, 2), 24 if "replicate_rows" in amg else 16
    cells = tuple(per * g for g in grid)
    length = tuple(c / float(cells[0]) for c in cells)
    params = {"eigensolver": {"number of eigenvectors": 2}, "agglomeration": {"nx": 2, "ny": 2, "nz": 2},
              "smoother": {"type": "Chebyshev", "degree": 3, "smoothing_range": 20.0},
              "solver": {"type": "amg", "amg": dict(amg)}, "is preconditioner": False}
    n_cycles = 20
    # ---- the single-process hierarchy on the global mesh
    gctx = M.Context()
    h_cell = tuple(length[d] / cells[d] for d in range(3))
    gprob = M.LaplaceProblem(cells, material, device="cuda", cell_size=h_cell)
    hg = M.Hierarchy(gctx, "HipMatrixFreeMeshEvaluator", gprob, params)
    deg, glmin, glmax = hg.smoother_info()
    ng = gprob.n_dofs
    rng = np.random.default_rng(0)
    x0g = np.where((gprob.constrained == 1).cpu().numpy(), 0.0, rng.random(ng))
    xs = torch.from_numpy(x0g).cuda(); bs = torch.zeros(ng, dtype=torch.float64, device="cuda"); rs = torch.empty_like(xs)
    hist_g = []
    for _ in range(n_cycles + 1):
        hg.operator_apply(0, xs, rs)
        hist_g.append(gctx.l2_norm(rs))
        hg.apply(bs, xs)
    hist_g = np.array(hist_g)

    # ---- eight ranks, one thread each
    n_ranks = 8
    mb = Mailboxes(n_ranks)
    hists, errors, info, overlapped = [None] * n_ranks, [None] * n_ranks, [None] * n_ranks, [0] * n_ranks
    x_final = np.zeros(ng)

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            part = M.BoxPartition(cells, rank, grid, length=length, low_ghost_cells=low_ghost)
            ctx = M.Context()
            tr = M.HaloTransport(ctx, part, callbacks=mb.callbacks(rank))
            assert tr.name() == "host" and tr.selftest(1024) == 0.0
            h = M.Hierarchy(ctx, "HipMatrixFreeMeshEvaluator", part.local_problem(material, "cuda"), params)
            _, lmin, lmax = h.smoother_info()
            assert abs(lmax - glmax) < 1e-9 * glmax and abs(lmin - glmin) < 1e-9 * glmax
            if material == "constant":
                assert h.smoother_sweep_terms() == (2, 3 if low_ghost == 4 else 0), h.smoother_sweep_terms()
            own_l, own_g, loc_g = (t.numpy() for t in tr.space_index(1))
            x = torch.from_numpy(x0g[loc_g]).cuda()
            # ghosts must come from the exchanges
            ghost = np.ones(len(loc_g), bool); ghost[own_l] = False
            x[torch.from_numpy(ghost).cuda()] = 1e30
            b0 = torch.zeros_like(x); r = torch.empty_like(x)
            v0 = tr.exchange_volume()
            h.operator_apply(0, x, r)
            assert tr.exchange_volume() - v0 == part.exchange_doubles()
            hist = []
            for _ in range(n_cycles + 1):
                h.operator_apply(0, x, r)
                hist.append(tr.owned_norm(r))
                h.apply(b0, x)
            hists[rank] = np.array(hist)
            x_final[own_g] = x.cpu().numpy()[own_l]
            info[rank] = (tr.n_exchanges(), tr.space(1)["n_spaces"], h.coarse_amg_gather_rows())
            ctx.synchronize()
            # ---- exchanges in flight on BOTH streams at once: behind the reflecting transport -- every message
            # this rank sends is mirrored on the device, nothing synchronises the host -- the ghost entries of b travel on the
            # exchange stream beside the pre-smoother while x is exchanged on the compute stream (per-stream staging buffers; a
            # shared one let the packed regions of the two overwrite each other).  With the overlap switched off every exchange
            # runs in program order on the compute stream: the same bits.  (The other ranks may still be in their cycles above: a
            # reflecting rank needs no partner.)
            tr.reflect()
            gl = torch.Generator(device="cuda").manual_seed(100 + rank)
            xa = torch.rand(len(loc_g), dtype=torch.float64, device="cuda", generator=gl)
            br = torch.rand(len(loc_g), dtype=torch.float64, device="cuda", generator=gl)
            xb = xa.clone()
            for xv, overlap in ((xa, True), (xb, False)):
                ctx.set_overlap_exchange(overlap)
                for _ in range(3):
                    h.apply(br, xv)
            ctx.set_overlap_exchange(True)
            ctx.synchronize()
            own_t = torch.from_numpy(own_l).cuda()
            assert torch.isfinite(xa[own_t]).all() and torch.equal(xa[own_t], xb[own_t]), "exchanges on two streams changed the result"
            overlapped[rank] = tr.n_overlapped()
        except BaseException as e:  # noqa: BLE001 - reported by the main thread
            errors[rank] = e
            mb.barrier.abort()
            for p in range(n_ranks):      # wake the neighbours that wait for a message of this rank
                mb.q[(rank, p)].put(np.zeros(0))

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(n_ranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=900)
    first = next((e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)), None) or \
        next((e for e in errors if e is not None), None)
    if first is not None:
        raise first
    assert all(h is not None for h in hists)
    if material == "constant":
        assert sum(overlapped) > 0      # (the ghost entries of b did travel on the second stream: the smoother sweep / the one-pass restriction read them)
    for hst in hists[1:]:
        np.testing.assert_array_equal(hst, hists[0])          # every rank sees the same (all-reduced) norms
    floor = 1e-12 * hist_g[0]
    np.testing.assert_allclose(hists[0], hist_g, rtol=1e-10, atol=floor)
    np.testing.assert_allclose(x_final, xs.cpu().numpy(), rtol=0, atol=1e-9 * np.abs(x0g).max())
    if "replicate_rows" in amg:
        # two aggregation levels stayed distributed along all three axes (spaces: local, fine, A_c, two levels), the third was
        # gathered through the permutation of the rank-ordered blocks
        # (+ the two-planes-deep fine space of the one-pass residual restriction where the material lets it run)
        assert info[0][1] == (6 if material == "constant" else 5) and 0 < info[0][2] < hg.level_size(1)
    # the oracle's restatement of the cycle, built from the single-process level matrices
    mesh = O.StructuredMesh(cells)
    mesh.h = h_cell
    mf = O.MatrixFreeLaplace(mesh, gprob.coefficient.cpu().numpy())
    p = O.ChebyshevParams(deg, glmax, glmin)
    smoother = lambda b, xx: O.chebyshev_smoother_apply(mf.vmult, mf.diagonal_inverse(), p, b, xx)
    ho = O.TwoLevelHierarchy(mf.vmult, smoother, hg.restrictor().to_scipy(),
                             O.amg_coarse_solver(hg.coarse_amg_levels(), 1, pre_smoothing_levels=amg.get("pre_smoothing_levels")), 1, False)
    res_o, rate, _ = O.vcycle_history(ho, mf.vmult, np.zeros(ng), x0g, n_cycles=n_cycles)
    res_o = np.array(res_o)
    np.testing.assert_allclose(hists[0] / hists[0][0], res_o[:n_cycles + 1] / res_o[0], rtol=1e-9, atol=1e-12)
    assert rate < 0.6
