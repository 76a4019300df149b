"""Which torch (not semseg_amd) device ops does a headline step still launch?  Prints every non-ss device kernel of one step with the
python line that launched it (torch.profiler with stacks)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
x, lab = bench.synthetic_batch(4, 512, 0, dev)
def step():
    flat.zero_grad(); out = net(x); loss = crit(out, lab); met.update(out, lab); loss.backward(); opt.step(red.wait())
for _ in range(3): step()
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode


class Log(TorchDispatchMode):
    """every aten op that reaches the dispatcher during one step, with the innermost frame of this repository that called it"""
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        base = name.split('.')[1] if '.' in name else name
        if base not in ('view', '_unsafe_view', 'reshape', 'permute', 'detach', 'alias', 'as_strided', 'slice', 'select', 'empty', 'empty_like', 'empty_strided', 'transpose', 'expand', 'unsqueeze', 'squeeze', 't', 'unbind', 'split', 'narrow', 'new_empty', 'lift_fresh', '_local_scalar_dense'):
            fr = [f for f in traceback.extract_stack() if ('semseg_amd' in f.filename or 'bench.py' in f.filename or 'tools/' in f.filename) and '__torch_dispatch__' not in f.name]
            where = f'{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].line[:70]}' if fr else '(autograd engine)'
            self.sites[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with Log() as log:
    step()
    torch.cuda.synchronize()
for (name, where), n in sorted(log.sites.items(), key=lambda kv: -kv[1]):
    print(f'{n:4d}  {name:34s} {where}')
