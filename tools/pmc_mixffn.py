"""Fused Mix-FFN launches at the stage-1 / stage-2 MiT-B2 shapes, for rocprofv3 --pmc passes (tools/pmc_passes.sh mixffn tools/pmc_mixffn.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd import ops
BF = torch.bfloat16
for (B, H, W, C) in [(16, 256, 256, 64), (16, 128, 128, 128)]:
    Ch, T = 4 * C, B * H * W
    r = lambda *s, sc=1.0: torch.randn(*s, device='cuda') * sc
    x, res, dout = r(T, C).to(BF), r(T, C).to(BF), r(T, C).to(BF)
    w1, w2 = r(Ch, C, sc=C ** -0.5).to(BF), r(C, Ch, sc=Ch ** -0.5).to(BF)
    b1, bd, b2, w9c = 0.1 * r(Ch), 0.1 * r(Ch), 0.1 * r(C), r(9, Ch, sc=0.3)
    keep = torch.ones(B, device='cuda')
    for _ in range(2):
        ops.mixffn_fwd(x, w1, b1, w9c, bd, w2, b2, B, H, W, res=res, keep=keep)
        if hasattr(ops, 'mixffn_bwd'):
            ops.mixffn_bwd(dout, x, w1, b1, w9c, bd, w2, B, H, W, keep=keep)
torch.cuda.synchronize()
print('done')
