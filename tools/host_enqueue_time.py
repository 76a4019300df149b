import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
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
B, S = int(os.environ.get("HB", 16)), int(os.environ.get("HS", 1024))
x, lab = bench.synthetic_batch(B, S, 0, dev)
def step():
    flat.zero_grad(); out = net(x); loss = crit(out, lab); met.update(out, lab); loss.backward(); opt.step(red.wait())
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'host enqueue {1e3*(t1-t0)/10:.1f} ms/step, total {1e3*(t2-t0)/10:.1f} ms/step')
