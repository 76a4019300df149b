"""Per-launch timeline of one TopFormer-B forward (B = 4, 2048 x 2048): entry point, key shape arguments, microseconds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd import _lib
from semseg_amd.topformer import TopFormerModel
from semseg_amd.detinit import det_fill_
from semseg_amd.runtime import set_compute_dtype

set_compute_dtype(torch.bfloat16)
head = sys.argv[1] if len(sys.argv) > 1 else 'simple'
net = det_fill_(TopFormerModel(6, 'B', head_cfg=frozenset({'alias': head}.items()))).cuda().eval()
x = torch.randn(4, 3, 2048, 2048, device='cuda')
for _ in range(3):
    net(x)
torch.cuda.synchronize()
_lib.TRACE = []
net(x)
torch.cuda.synchronize()
trace, _lib.TRACE = _lib.TRACE, None
tot = 0.0
for name, a, e0, e1 in trace:
    us = e0.elapsed_time(e1) * 1e3
    tot += us
    ints = [v for v in a if isinstance(v, int) and 0 < v < (1 << 24)]
    print(f'{name:24s} {us:8.1f} us  {ints[:12]}')
print(f'total {tot:.1f} us over {len(trace)} launches')
