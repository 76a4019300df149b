"""Encoder-only timing probe (NOT the headline bench): MiT fwd+bwd at a given batch/size/dtype."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, time, torch
from semseg_amd.segformer import SegFormer
from semseg_amd.detinit import det_fill_
from semseg_amd.runtime import set_compute_dtype

ap = argparse.ArgumentParser()
ap.add_argument('--variant', default='B2'); ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--size', type=int, default=1024); ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--dtype', default='bf16')
a = ap.parse_args()
set_compute_dtype(torch.bfloat16 if a.dtype == 'bf16' else torch.float32)
enc = det_fill_(SegFormer(a.variant)).cuda().train()
x = torch.rand(a.batch, 3, a.size, a.size, device='cuda')
for i in range(a.steps + 1):
    if i == 1:
        torch.cuda.synchronize(); t0 = time.time()
    outs = enc(x)
    loss = sum(o.float().mean() for o in outs)
    loss.backward()
    enc.zero_grad(set_to_none=True)
torch.cuda.synchronize()
dt = (time.time() - t0) / a.steps
print(f'{a.variant} B={a.batch} S={a.size} {a.dtype}: {dt*1e3:.1f} ms/step, {a.batch/dt:.1f} img/s, peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB')
