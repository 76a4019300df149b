"""Stride-2 depthwise forward / data gradient / weight gradient at the TopFormer-B training shapes (B = 16, 1024^2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd import ops


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B = 16
for (H, C, k, pt) in [(512, 64, 3, 0), (256, 96, 5, 1), (128, 192, 3, 0), (64, 768, 5, 1)]:
    Ho = H // 2
    x = torch.randn(B * H * H, C, device='cuda').bfloat16()
    w = torch.randn(k * k, C, device='cuda') * 0.2
    g = torch.randn(B * Ho * Ho, C, device='cuda').bfloat16()
    line = f'{H}->{Ho} C={C} k={k}:'
    line += f'  fwd {timeit(lambda: ops.dwconvk_fwd(x, w, None, B, H, H, k, 2, pt, pt, Ho, Ho, act=3)):7.1f} us'
    line += f'  dgrad {timeit(lambda: ops.dwconvk_dgrad(g, w, B, H, H, k, 2, pt, pt, Ho, Ho)):7.1f} us'
    line += f'  wgrad {timeit(lambda: ops.dwconvk_wgrad_s(g, x, B, H, H, k, 2, pt, pt, Ho, Ho)):7.1f} us'
    mb = (x.numel() + g.numel()) * 2 / 1e6
    print(line + f'   ({mb:.0f} MB in + out)', flush=True)
