"""dwconv3 fwd (GELU / plain), two-pass bwd and fused bwd launches on the stage-1 MiT-B2 shape, for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd import ops
B, H, W, C = 16, 256, 256, 256
x = torch.randn(B * H * W, C, device='cuda').bfloat16()
dy = torch.randn(B * H * W, C, device='cuda').bfloat16()
w = torch.randn(9, C, device='cuda') / 3
b = torch.randn(C, device='cuda') * 0.1
for _ in range(2):
    ops.dwconv3_fwd(x, w, b, B, H, W, act=1)
    ops.dwconv3_fwd(x, w, b, B, H, W, act=0)
    ops.dwconv3_bwd(dy, x, w, b, B, H, W, act=1)
    ops.dwconv3_bwd_fused(dy, x, w, b, B, H, W, act=1)
torch.cuda.synchronize()
print('done')
