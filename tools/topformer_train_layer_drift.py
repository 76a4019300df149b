"""Where does a bf16 (or perturbed-input) train-mode forward of TopFormer leave the fp32 one?  Records every tape tensor of
topformer_train.py in creation order for two runs and prints the relative L2 distance per tensor."""
import sys
import torch
sys.path.insert(0, '.')
from semseg_amd.detinit import det_fill_, det_input
from semseg_amd.topformer import TopFormerModel
from semseg_amd.runtime import set_compute_dtype
from semseg_amd import topformer_train as tt

shape = tuple(int(v) for v in sys.argv[1].split('x')) if len(sys.argv) > 1 else (2, 3, 256, 256)
mode = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
log = []
orig = tt._V.__init__


def rec(self, t):
    orig(self, t)
    log.append(t)
tt._V.__init__ = rec
net = det_fill_(TopFormerModel(6, 'T')).cuda().train()
x = det_input('top.xt', shape).cuda()
with torch.no_grad():
    net(x)
    a, log = log, []
    tt._V.__init__ = rec
    if mode == 'bf16':
        set_compute_dtype(torch.bfloat16)
        net(x)
        set_compute_dtype(None)
    else:
        g = torch.Generator().manual_seed(1)
        net(x * (1 + float(mode) * torch.randn(x.shape, generator=g).cuda()))
    b = log
for i, (u, v) in enumerate(zip(a, b)):
    u, v = u.float(), v.float()
    d = float((u - v).norm() / max(float(u.norm()), 1e-30))
    var = u.var(0)
    print(f'{i:3d} {tuple(u.shape)!s:16s} rel-L2 {d:9.2e}   |x| {float(u.abs().mean()):9.2e}  min/median channel variance {float(var.min()):9.2e} {float(var.median()):9.2e}')
