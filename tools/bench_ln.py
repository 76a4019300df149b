"""LayerNorm forward / backward at the MiT-B2 stage shapes of the headline step (bf16): GB/s of algorithmic bytes."""
import sys
import torch
sys.path.insert(0, '.')
from semseg_amd import ops

T = torch.bfloat16


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rows, C in ((1 << 20, 64), (1 << 18, 128), (1 << 16, 320), (1 << 14, 512), (1 << 18, 512)):
    x = torch.randn(rows, C, device='cuda').to(T)
    dy = torch.randn(rows, C, device='cuda').to(T)
    res = torch.randn(rows, C, device='cuda').to(T)
    g, b = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5)
    tf = timed(lambda: ops.layernorm_fwd(x, g, b, 1e-5))
    tb = timed(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, res=res))
    byt = rows * C * 2
    print(f'rows {rows:8d} C {C:4d}  fwd {tf:7.1f} us {2 * byt / tf / 1e3:7.0f} GB/s   bwd(+res) {tb:7.1f} us {4 * byt / tb / 1e3:7.0f} GB/s')
