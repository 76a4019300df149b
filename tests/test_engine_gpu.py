"""Gradient delivery into engine.FlatParams' flat buffer (runtime.deliver_grads + ss_multi_add_f32) against plain autograd."""
import pytest
import torch

from semseg_amd.detinit import det_fill_, det_input, det_labels

pytestmark = pytest.mark.gpu


def _step(net, crit, x, lab):
    loss = crit(net(x), lab)
    loss.backward()
    return loss


def test_flat_gradients_equal_autograd_gradients():
    from semseg_amd.model import SegFormerLawin
    from semseg_amd.loss import OhemCrossEntropyLoss
    from semseg_amd.engine import FlatParams, AdaBound
    net = SegFormerLawin('B0', 128, 6)
    det_fill_(net.encoder), det_fill_(net.decoder)
    net = net.cuda().eval()                                   # eval: no DropPath / Dropout draws, the two passes see the same graph
    x, lab = det_input('eng.x', (2, 3, 64, 64)).cuda(), det_labels('eng.lab', (2, 64, 64), 6, cell=8).cuda()
    _step(net, OhemCrossEntropyLoss(0, 0.7, 0.1), x, lab)
    ref = {n: p.grad.clone() for n, p in net.named_parameters()}
    assert all(g is not None for g in ref.values())
    for p in net.parameters():
        p.grad = None
    flat = FlatParams(net)
    fired = []
    for p in flat.params:
        p._ss_grad_ready = fired.append
    flat.zero_grad()
    _step(net, OhemCrossEntropyLoss(0, 0.7, 0.1), x, lab)
    for n, p in net.named_parameters():
        assert p.grad.data_ptr() == p._ss_grad_sink.data_ptr()                  # still the view of the flat buffer
        assert torch.equal(p.grad, ref[n]), n
    assert len(fired) == len(flat.params) and {id(p) for p in fired} == {id(p) for p in flat.params}
    _step(net, OhemCrossEntropyLoss(0, 0.7, 0.1), x, lab)                       # no zero_grad in between: accumulates like autograd
    for n, p in net.named_parameters():
        assert torch.allclose(p.grad, 2 * ref[n], rtol=1e-6, atol=0), n
    # the flat optimizer consumes the same buffer
    before = flat.data.clone()
    AdaBound(flat).step()
    assert not torch.equal(before, flat.data)


def test_multi_add_ragged_sizes():
    from semseg_amd import ops
    torch.manual_seed(0)
    sizes = [1, 3, 4, 4095, 4096, 4097, 70000, 5] * 9                           # > 48 tensors: two launches; odd sizes: scalar tails
    pool = torch.randn(sum(sizes) + 8, device='cuda')
    srcs, dsts, o = [], [], 1                                                   # offset 1: unaligned slots
    for n in sizes:
        srcs.append(torch.randn(n, device='cuda'))
        dsts.append(pool[o:o + n])
        o += n
    slab = torch.randn(5, 64, 65, device='cuda')                                # strided sources: [64][64] rows 65 apart + the 65th column
    for h in range(5):
        for src in (slab[h, :, :64], slab[h, :, 64]):
            srcs.append(src)
            dsts.append(torch.randn(src.numel(), device='cuda'))
    want = [d + s.reshape(-1) for s, d in zip(srcs, dsts)]
    edge = (pool[0].item(), pool[o:].clone())
    ops.multi_add(srcs, dsts)
    for d, w in zip(dsts, want):
        assert torch.equal(d, w)
    assert pool[0].item() == edge[0] and torch.equal(pool[o:], edge[1])
