import torch
a = torch.empty(1 << 28, dtype=torch.float64, device='cuda').normal_()   # 2 GiB
c = torch.empty_like(a)
def t(f, n=10):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
ms = t(lambda: c.copy_(a)); print(f"copy 2GiB->2GiB  {ms:.3f} ms  {2*a.numel()*8/ms/1e6:.0f} GB/s")
ms = t(lambda: a.sum());    print(f"sum 2GiB         {ms:.3f} ms  {a.numel()*8/ms/1e6:.0f} GB/s")
ms = t(lambda: torch.add(a, c, out=c)); print(f"add (2 reads 1 write) {ms:.3f} ms  {3*a.numel()*8/ms/1e6:.0f} GB/s")
b = a[: 1 << 24]; d = c[: 1 << 24]   # 128 MiB vectors (the 257^3 scale)
ms = t(lambda: d.copy_(b), 50); print(f"copy 128MiB  {ms:.4f} ms  {2*b.numel()*8/ms/1e6:.0f} GB/s")
