"""MiT attention forward / backward at the stage-1 and stage-3 shapes of the headline step, for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd import ops
B = 16
for N, heads in [(65536, 1), (4096, 5)]:
    C, Nk = heads * 64, 1024
    q = torch.randn(B * N, C, device='cuda').bfloat16(); kv = torch.randn(B * Nk, 2 * C, device='cuda').bfloat16()
    do = torch.randn(B * N, C, device='cuda').bfloat16()
    for _ in range(2):
        o, lse = ops.mit_attn_fwd(q, kv, B, N, Nk, heads)
        ops.mit_attn_bwd(q, kv, o, do, lse, B, N, Nk, heads)
    torch.cuda.synchronize()
    break
print('done')
