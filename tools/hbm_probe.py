"""GPU box: what this MI355X reaches on plain streams (torch kernels), to put the kernels' fractions of the 8 TB/s spec peak in
context: read-only (sum), copy (read + write), write-only (fill).  python tools/hbm_probe.py"""
import torch, time
dev = "cuda"
def bw(fn, nbytes, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return nbytes * reps / (time.perf_counter() - t0) / 1e12
for mb in (128, 1024, 4096):
    n = mb * (1 << 20) // 4
    x = torch.ones(n, dtype=torch.float32, device=dev); y = torch.empty_like(x)
    print(f"{mb:5d} MiB  read-only sum {bw(lambda: x.sum(), 4*n):.2f} TB/s   copy {bw(lambda: y.copy_(x), 8*n):.2f} TB/s (read+write)   fill {bw(lambda: y.fill_(1.0), 4*n):.2f} TB/s")
    del x, y
