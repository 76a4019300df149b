"""Per-gradient bf16 noise of the DNet training step against the fp32 reference golden, beside the reference's OWN bf16-autocast deviation
(tests/golden/dnet_x_128.npz): ours / reference ratio per tensor.  SEMSEG_DNET_PROBE selects experiments (see semseg_amd)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from semseg_amd.decoder import DNet
from semseg_amd.loss import DiceLoss
from semseg_amd.runtime import set_compute_dtype
from semseg_amd.detinit import det_fill_, det_input, det_labels

g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'dnet_x_128.npz'))
net = det_fill_(DNet(3, 6, embed_dim=128, stem_code='X')).cuda().train()
net.decoder.decoder.dropout.p = 0.0
x = det_input('dnet.x', (2, 3, 128, 128)).cuda()
lab = det_labels('dnet.lab', (2, 128, 128), 6, cell=8).cuda()
set_compute_dtype(torch.bfloat16)
probs = net(x)
loss = DiceLoss(0.5, False, 6)(probs, lab)
loss.backward()
set_compute_dtype(None)
l2 = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
print(f'probs rel-L2 {l2(probs.detach().float().cpu().numpy(), g["probs"]):.3e}   (reference bf16: {l2(g["bf16.probs"], g["probs"]) if "bf16.probs" in g.files else float("nan"):.3e})')
params = dict(net.named_parameters())
rows = []
for k in g.files:
    if k.startswith('grad.'):
        ours = l2(params[k[5:]].grad.float().cpu().numpy(), g[k])
        ref = l2(g['bf16.' + k], g[k])
        rows.append((ours / ref, ours, ref, k[5:]))
for r, o, f, k in sorted(rows, reverse=True):
    print(f'{r:6.2f}x  ours {o:.3e}  reference-bf16 {f:.3e}  {k}')
