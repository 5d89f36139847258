# Diagnosis of the 512^3 smoother leg: per-apply times from a cold start, vector
# placement (offsets inside one allocation), 512 vs 513 DoFs per direction, both materials.
# usage: diag512.py [material]
import os, sys, time, json, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import mfmg_amd as M
from bench import smoother_coefficients

material = sys.argv[1] if len(sys.argv) > 1 else 'constant'
ctx = M.Context()
coefs = smoother_coefficients(3, 0.09, 1.8)


def sweep(op, b, x, s1, s2):
    op.smoother_step(b, x, None, coefs[0][0], coefs[0][1], s2)
    op.smoother_step(b, s2, x, coefs[1][0], coefs[1][1], s1)
    op.smoother_step(b, s1, s2, coefs[2][0], coefs[2][1], x)


def timed(op, vecs, reps, warm):
    b, x, s1, s2 = vecs
    for _ in range(warm):
        sweep(op, b, x, s1, s2)
    ctx.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for r in range(reps):
        sweep(op, b, x, s1, s2)
        ev[r + 1].record()
    ev[-1].synchronize()
    return [ev[r].elapsed_time(ev[r + 1]) for r in range(reps)]


def stats(ts):
    s = sorted(ts)
    return {"min": round(s[0], 3), "median": round(s[len(s) // 2], 3), "max": round(s[-1], 3),
            "first5": [round(v, 3) for v in ts[:5]]}


for ndof in (512, 513):
    prob = M.LaplaceProblem((ndof - 1,) * 3, material, device='cuda')
    op = M.MatrixFreeLaplace(ctx, prob)
    N = prob.n_dofs
    del prob
    torch.cuda.empty_cache()
    print(ndof, 'tile', op.get_tile() if hasattr(op, 'get_tile') else None, flush=True)
    g = torch.Generator(device='cuda').manual_seed(1)
    # (a) four separate torch allocations, as bench.py does
    x = torch.rand(N, dtype=torch.float64, device='cuda', generator=g)
    b = torch.zeros(N, dtype=torch.float64, device='cuda')
    s1, s2 = torch.empty_like(x), torch.empty_like(x)
    print(ndof, 'ptrs mod 2MiB', [hex(t.data_ptr() % (1 << 21)) for t in (b, x, s1, s2)], flush=True)
    time.sleep(3.0)   # idle GPU first: does a cold start cost?
    print(json.dumps({"n": ndof, "case": "separate allocations, cold, no warm-up", **stats(timed(op, (b, x, s1, s2), 20, 0))}), flush=True)
    print(json.dumps({"n": ndof, "case": "separate allocations, warm", **stats(timed(op, (b, x, s1, s2), 20, 3))}), flush=True)
    del x, b, s1, s2
    torch.cuda.empty_cache()
    # (b) one allocation, vectors offset by different odd multiples of 4 KiB
    for pad_kib in (0, 4, 12, 68, 260):
        pad = pad_kib * 128   # doubles
        stride = N + pad
        big = torch.zeros(4 * stride + 8 * pad + 1024, dtype=torch.float64, device='cuda')
        vecs = []
        for q, mult in enumerate((0, 1, 3, 5)):
            off = q * stride + mult * pad
            off = (off + 1) // 2 * 2     # 16-byte aligned
            vecs.append(big[off:off + N])
        vecs[1].copy_(torch.rand(N, dtype=torch.float64, device='cuda', generator=g))
        print(json.dumps({"n": ndof, "case": f"one allocation, pad {pad_kib} KiB x (0,1,3,5)", **stats(timed(op, vecs, 12, 2))}), flush=True)
        del vecs, big
        torch.cuda.empty_cache()
    del op
    torch.cuda.empty_cache()
