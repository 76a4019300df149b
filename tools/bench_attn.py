"""Timings of the MiT attention forward / backward at the four stage shapes of the headline step (B=16, 1024^2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd import ops
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B = 16
for N, heads in [(65536, 1), (16384, 2), (4096, 5), (1024, 8)]:
    C, Nk = heads * 64, 1024
    q = torch.randn(B * N, C, device='cuda').bfloat16(); kv = torch.randn(B * Nk, 2 * C, device='cuda').bfloat16()
    do = torch.randn(B * N, C, device='cuda').bfloat16()
    o, lse = ops.mit_attn_fwd(q, kv, B, N, Nk, heads)
    tf = timeit(lambda: ops.mit_attn_fwd(q, kv, B, N, Nk, heads))
    tb = timeit(lambda: ops.mit_attn_bwd(q, kv, o, do, lse, B, N, Nk, heads))
    fl = 4.0 * B * heads * N * Nk * 64
    print(f'wg={os.environ.get("SS_ATTN_WG", "1024")} N={N:6d} heads={heads} fwd {tf*1e3:7.1f} us {fl/tf/1e9:6.0f} TF/s   bwd {tb*1e3:7.1f} us {2.5*fl/tb/1e9:6.0f} TF/s', flush=True)
