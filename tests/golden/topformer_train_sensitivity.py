"""Sensitivity of the REFERENCE's own train-mode forward (CPU fp32) to a one-ulp change of the input image, at the golden size:
    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/topformer_train_sensitivity.py B:refiner
The figure is the floor to which two fp32 implementations of this step can agree (recorded in tests/test_topformer_train_gpu.py)."""
import os
import sys
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(1, '/root/reference')
sys.dont_write_bytecode = True
from semseg_amd.detinit import det_fill_, det_input  # noqa: E402
from core.models import TopFormerModel  # noqa: E402  (reference)

alias, head = sys.argv[1].split(':')
net = det_fill_(TopFormerModel(6, alias, 3, 'dot_sum', 'sum', frozenset({'alias': head}.items()))).train()
x = det_input('top.xt', (2, 3, 128, 192))
with torch.no_grad():
    a = net(x)
    for eps in (6e-8, 1e-6):
        g = torch.Generator().manual_seed(1)
        b = net(x * (1 + eps * torch.randn(x.shape, generator=g)))
        print(alias, head, 'reference, train mode: input moved by', eps, '-> logits moved by', float((a - b).abs().max() / a.abs().max()))
    a64 = net.double()(x.double())
    print(alias, head, 'reference fp32 vs the reference in fp64:', float((a.double() - a64).abs().max() / a64.abs().max()))
