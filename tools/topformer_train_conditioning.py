"""How well conditioned is one train-mode forward of TopFormer at the golden size (2 x 3 x 128 x 192)?  The transformer part sees only
2 * 2 * 3 = 12 token rows there, so its BatchNorms divide by the spread of 12 samples.  Prints the relative change of the logits when the
input image moves by one fp32 ulp (relative 6e-8): the floor any two fp32 implementations can agree to."""
import sys
import torch
sys.path.insert(0, '.')
from semseg_amd.detinit import det_fill_, det_input
from semseg_amd.topformer import TopFormerModel

from semseg_amd.runtime import set_compute_dtype
shape = tuple(int(v) for v in sys.argv[1].split('x')) if len(sys.argv) > 1 else (2, 3, 128, 192)
for alias, head in (('T', 'simple'), ('B', 'simple'), ('B', 'refiner')):
    net = det_fill_(TopFormerModel(6, alias, head_cfg=frozenset({'alias': head}.items()))).cuda().train()
    x = det_input('top.xt', shape).cuda()
    with torch.no_grad():
        a = net(x).float()
        set_compute_dtype(torch.bfloat16)
        b = net(x).float()
        set_compute_dtype(None)
        print(alias, head, shape, 'train mode: bf16 vs fp32 logits rel-L2', float((a - b).norm() / a.norm()))
    with torch.no_grad():
        a = net(x).float()
        for eps in (6e-8, 1e-6):
            g = torch.Generator().manual_seed(1)
            b = net(x * (1 + eps * torch.randn(x.shape, generator=g).cuda())).float()
            print(alias, head, 'input moved by', eps, '-> logits moved by', float((a - b).abs().max() / a.abs().max()))
        net.eval()
        a = net(x).float()
        g = torch.Generator().manual_seed(1)
        b = net(x * (1 + 6e-8 * torch.randn(x.shape, generator=g).cuda())).float()
        print(alias, head, 'eval mode, input moved by 6e-8 -> logits moved by', float((a - b).abs().max() / a.abs().max()))
