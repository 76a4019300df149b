"""Mix-FFN depthwise 3x3 backward at the headline shapes: two-pass (ss_dwconv3_bwd + flipped ss_dwconv3_fwd) vs ss_dwconv3_bwd_fused."""
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

for (B, H, W, C) in [(16, 256, 256, 256), (16, 128, 128, 512), (16, 64, 64, 1280), (16, 32, 32, 2048), (2, 512, 512, 256)]:
    x = torch.randn(B * H * W, C, device='cuda').bfloat16()
    dy = torch.randn(B * H * W, C, device='cuda').bfloat16()
    w = torch.randn(9, C, device='cuda') / 3
    wf = w.flip(0).contiguous()
    b = torch.randn(C, device='cuda') * 0.1
    def two():
        g, dwdb = ops.dwconv3_bwd(dy, x, w, b, B, H, W, act=1)
        return ops.dwconv3_fwd(g, wf, None, B, H, W, act=0), dwdb
    def one():
        return ops.dwconv3_bwd_fused(dy, x, w, b, B, H, W, act=1)
    a, bq = two(), one()
    err = float((a[0].float() - bq[0].float()).abs().max() / a[0].float().abs().max())
    errw = float((a[1] - bq[1]).abs().max() / a[1].abs().max())
    t2, t1 = t(two), t(one)
    byt = 3 * B * H * W * C * 2
    print(f'{B}x{H}x{W}x{C}: two-pass {t2:7.1f} us  fused {t1:7.1f} us ({byt / t1 / 1e6:5.2f} TB/s)  dx diff {err:.1e} dW diff {errw:.1e}', flush=True)
