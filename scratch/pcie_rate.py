# host <-> device copy rates on the box: pageable vs pinned host memory
import time, torch
n = 1 << 28  # 2 GiB of doubles
d = torch.empty(n, dtype=torch.float64, device="cuda")
for pin in (False, True):
    h = torch.empty(n, dtype=torch.float64, pin_memory=pin); h.fill_(1.0)
    for name, f in (("H2D", lambda: d.copy_(h)), ("D2H", lambda: h.copy_(d))):
        f(); torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f"{'pinned' if pin else 'pageable'} {name}: {n * 8 / dt / 1e9:.1f} GB/s")
t = time.perf_counter(); h2 = torch.empty(n, dtype=torch.float64); h2.fill_(0.0); print(f"allocate + first touch of 2 GiB: {time.perf_counter() - t:.2f} s")
