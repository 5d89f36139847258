"""Host cost of one halo exchange through torch.distributed on RCCL, measured on a one-rank group with
send/recv to self (the only peer a one-GPU box has): time on the calling thread per batch_isend_irecv + wait."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
w = torch.zeros(1, device="cuda"); dist.all_reduce(w); torch.cuda.synchronize()
n = 257 * 257
s = torch.rand(n, dtype=torch.float64, device="cuda"); r = torch.zeros_like(s)
side = torch.cuda.Stream()
try:
    ops = [dist.P2POp(dist.isend, s, 0), dist.P2POp(dist.irecv, r, 0)]
    for it in range(3):
        with torch.cuda.stream(side):
            for q in dist.batch_isend_irecv(ops): q.wait()
    torch.cuda.synchronize()
    assert torch.equal(s, r)
    t0 = time.perf_counter(); K = 200
    for it in range(K):
        with torch.cuda.stream(side):
            for q in dist.batch_isend_irecv(ops): q.wait()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host per exchange {(t1 - t0) / K * 1e6:.1f} us, incl. device {(t2 - t0) / K * 1e6:.1f} us")
except Exception as e:
    print("self p2p failed:", repr(e))
dist.destroy_process_group()
