"""Mix-FFN depthwise 3x3 + GELU forward at the headline shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd import ops

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (B, H, W, C) in [(16, 256, 256, 256), (16, 128, 128, 512), (16, 64, 64, 1280), (16, 32, 32, 2048)]:
    x = torch.randn(B * H * W, C, device='cuda').bfloat16()
    w = torch.randn(9, C, device='cuda') / 3
    b = torch.randn(C, device='cuda') * 0.1
    us = t(lambda: ops.dwconv3_fwd(x, w, b, B, H, W, act=1))
    print(f'{B}x{H}x{W}x{C}: {us:7.1f} us ({2 * B * H * W * C * 2 / us / 1e6:5.2f} TB/s)', flush=True)
