"""Which torch (not semseg_amd) device ops does a headline step still launch?  Prints copy_/fill_/cat/... counts by calling line."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from semseg_amd.model import SegFormerLawin
from semseg_amd.loss import OhemCrossEntropyLoss
from semseg_amd.metrics import SegmentationMetrics
from semseg_amd.detinit import det_fill_
from semseg_amd.runtime import set_compute_dtype
from semseg_amd.engine import FlatParams, GradReducer, AdaBound
import bench
set_compute_dtype(torch.bfloat16)
dev = torch.device('cuda', 0)
net = SegFormerLawin('B2', 512, 6); det_fill_(net.encoder), det_fill_(net.decoder)
net = net.to(dev).train()
flat = FlatParams(net); red = GradReducer(flat); opt = AdaBound(flat, lr=1e-3, final_lr=0.1)
crit = OhemCrossEntropyLoss(0, 0.7, 0.1); met = SegmentationMetrics(6, ignore_index=0).to(dev)
x, lab = bench.synthetic_batch(4, 512, 0, dev)
def step():
    flat.zero_grad(); out = net(x); loss = crit(out, lab); met.update(out, lab); loss.backward(); opt.step(red.wait())
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.count)
for e in rows[:45]:
    print(f'{e.count:5d}  {e.key[:100]}')
