"""Achievable HBM rates of plain streams on this box: write-only (fill), read+write (copy), read-only (sum)."""
import torch
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for gib in (0.25, 1, 4):
    n = int(gib * 2 ** 30) // 2
    a = torch.empty(n, dtype=torch.bfloat16, device='cuda'); b = torch.empty_like(a)
    tz = timeit(lambda: a.zero_()); tc = timeit(lambda: b.copy_(a)); tr = timeit(lambda: a.view(torch.int32).sum())
    print(f'{gib:5.2f} GiB  fill {2*n/tz/1e6:7.0f} GB/s   copy {4*n/tc/1e6:7.0f} GB/s (read+write)   sum {2*n/tr/1e6:7.0f} GB/s', flush=True)
