# the multi-term smoother sweep against the term-by-term kernel, bit for bit (GPU)
import itertools, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import mfmg_amd as M
from bench import smoother_coefficients

def run(n, material, K, tile, dtype=torch.float64, stored=False):
    ctx = M.Context()
    if stored:
        ctx.set_stored_diagonal(True)
    prob = M.LaplaceProblem(n, 'constant' if material == 'cellwise' else material, device='cuda')
    if material == 'cellwise':  # one value per cell, varying from cell to cell
        gg = torch.Generator(device='cuda'); gg.manual_seed(7)
        prob.coefficient = (0.5 + torch.rand(prob.n_cells_total, 1, dtype=torch.float64, device='cuda', generator=gg)).expand(-1, 8).contiguous()
    op = M.MatrixFreeLaplace(ctx, prob) if dtype == torch.float64 else M.MatrixFreeLaplaceF32(ctx, prob)
    N = prob.n_dofs
    if not op.sweep_available(K):
        return 'n/a'
    g = torch.Generator(device='cuda'); g.manual_seed(1234)
    x = torch.rand(N, dtype=dtype, device='cuda', generator=g)
    b = torch.rand(N, dtype=dtype, device='cuda', generator=g)
    coefs = smoother_coefficients(3, 0.09, 1.8)
    al = [c[0] for c in coefs][:K]; be = [c[1] for c in coefs][:K]
    al[0] = 0.0
    its = [x]
    for k in range(K):
        o = torch.full_like(x, float('nan'))
        op.smoother_step(b, its[-1], its[-2] if k > 0 else None, al[k], be[k], o)
        its.append(o)
    ctx.synchronize()
    if tile is not None and dtype == torch.float64:
        op.set_sweep_tile(*tile)
    out = torch.full_like(x, float('nan')); outp = torch.full_like(x, float('nan'))
    op.smoother_sweep(al, be, b, x, out, outp)
    ctx.synchronize()
    ok = torch.equal(out, its[-1]) and torch.equal(outp, its[-2])
    if not ok:
        d = (out - its[-1]).abs(); dn = torch.isnan(out).sum().item()
        d2 = (outp - its[-2]).abs()
        bad = torch.nonzero(~(out == its[-1])).flatten()
        N0, N1 = n[0] + 1, n[1] + 1
        first = [(int(i) % N0, (int(i) // N0) % N1, int(i) // (N0 * N1)) for i in bad[:6]]
        return f'MISMATCH nan={dn} max={torch.nan_to_num(d).max().item():.3e} prev={torch.nan_to_num(d2).max().item():.3e} nbad={bad.numel()} first={first}'
    return 'ok'

cases = [((6, 5, 7), 'constant'), ((20, 17, 9), 'cellwise'), ((70, 30, 20), 'constant'), ((130, 40, 33), 'cellwise'), ((64, 64, 64), 'constant')]
tiles = [None, (4, 3, 8), (8, 3, 5), (2, 4, 7), (8, 2, 64), (1, 4, 3)]
fail = 0
for (n, mat), K, tile in itertools.product(cases, (2, 3), tiles):
    if tile is not None and tile[0] * tile[1] - 2 * K + 1 < 1:
        continue
    r = run(n, mat, K, tile)
    print(n, mat, 'K', K, 'tile', tile, r, flush=True)
    fail += r not in ('ok',)
r = run((40, 33, 21), 'constant', 3, (4, 3, 6), stored=True); print('stored diagonal', r); fail += r != 'ok'
r = run((40, 33, 21), 'cellwise', 3, None, dtype=torch.float32); print('fp32', r); fail += r != 'ok'
print('FAILURES', fail)
if len(sys.argv) > 1:
    # timing at 257^3
    ctx = M.Context()
    nn = int(sys.argv[1])
    prob = M.LaplaceProblem((nn - 1,) * 3, 'constant', device='cuda')
    op = M.MatrixFreeLaplace(ctx, prob)
    N = prob.n_dofs
    x = torch.rand(N, dtype=torch.float64, device='cuda'); b = torch.rand_like(x); o = torch.empty_like(x); o2 = torch.empty_like(x)
    coefs = smoother_coefficients(3, 0.09, 1.8)
    al = [0.0] + [c[0] for c in coefs][1:]; be = [c[1] for c in coefs]
    def timeit(f, reps=10):
        f(); ctx.synchronize()
        ctx.synchronize(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            f()
        ctx.synchronize()
        return (time.perf_counter() - t) / reps * 1e3
    def unfused():
        op.smoother_step(b, x, None, al[0], be[0], o)
        op.smoother_step(b, o, x, al[1], be[1], o2)
        op.smoother_step(b, o2, o, al[2], be[2], x)
    print(f'{nn}^3 unfused 3 terms: {timeit(unfused):.3f} ms', flush=True)
    tl = [(8, 3, 0), (8, 3, 37), (8, 3, 26), (4, 3, 0), (4, 3, 32), (8, 2, 0), (4, 4, 0), (8, 3, 64), (8, 3, 16)]
    for K in (3, 2):
        for tile in tl:
            if tile[0] * tile[1] - 2 * K + 1 < 1:
                continue
            op.set_sweep_tile(*tile)
            try:
                t = timeit(lambda: op.smoother_sweep(al[:K], be[:K], b, x, o, None))
            except Exception as e:
                print('tile', tile, 'failed', e); continue
            print(f'{nn}^3 sweep K={K} tile {tile} -> {op.get_sweep_tile(K)}: {t:.3f} ms', flush=True)
