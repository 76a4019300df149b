"""Timings of the depthwise k x k kernels at the ConvNeXt-T stage shapes of the DNet step (B = 8, 1024^2)."""
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
B, k = 8, 7
for S, C in [(256, 96), (128, 192), (64, 384), (32, 768)]:
    x = torch.randn(B * S * S, C, device='cuda').bfloat16(); g = torch.randn_like(x)
    w = torch.randn(k * k, C, device='cuda'); b = torch.randn(C, device='cuda')
    tf = timeit(lambda: ops.dwconvk_fwd(x, w, b, B, S, S, k))
    tw = timeit(lambda: ops.dwconvk_wgrad(g, x, B, S, S, k))
    n = B * S * S * C
    print(f'S={S:4d} C={C:4d} fwd {tf*1e3:7.1f} us ({4*n/tf/1e6:6.0f} GB/s, {2*49*n/tf/1e9:5.1f} TFLOP/s)   wgrad {tw*1e3:7.1f} us ({4*n/tw/1e6:6.0f} GB/s)', flush=True)
