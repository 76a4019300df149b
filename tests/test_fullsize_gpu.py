"""BASELINE-size checks of the headline network (MiT-B2 + Lawin(512), 1024 x 1024, bf16) through properties that do not need an
oracle run at that size: an image's logits do not depend on what else is in the batch (eval mode: every kernel reduces over
an image's own rows in an order fixed by the tile it sits in), and one training step leaves a finite, non-zero gradient
in every parameter."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net():
    from semseg_amd.model import SegFormerLawin
    from semseg_amd.detinit import det_fill_
    net = SegFormerLawin('B2', 512, 6)
    det_fill_(net.encoder), det_fill_(net.decoder)
    return net.cuda()


def test_full_size_eval_logits_are_batch_independent():
    from semseg_amd.runtime import set_compute_dtype
    from semseg_amd.detinit import det_input
    net = _net().eval()
    x = det_input('full.x', (2, 3, 1024, 1024), 0, 1).cuda()
    set_compute_dtype(torch.bfloat16)
    try:
        with torch.no_grad():
            both = net(x).float()
            one = [net(x[i:i + 1]).float() for i in range(2)]
    finally:
        set_compute_dtype(None)
    assert both.shape == (2, 6, 1024, 1024) and torch.isfinite(both).all()
    scale = float(both.abs().max())
    for i in range(2):
        # identical up to the split-K plan of a GEMM whose row count changed (bf16 rounding of one partial sum)
        assert float((both[i] - one[i][0]).abs().max()) <= 2.0 ** -7 * scale, i
    assert float((both[0] - both[1]).abs().max()) > 1e-3 * scale           # the two images differ, so the check is not vacuous


def test_full_size_training_step_reaches_every_parameter():
    from semseg_amd.runtime import set_compute_dtype
    from semseg_amd.detinit import det_input, det_labels
    from semseg_amd.loss import OhemCrossEntropyLoss
    net = _net().train()
    x = det_input('full.x', (2, 3, 1024, 1024), 0, 1).cuda()
    lab = det_labels('full.lab', (2, 1024, 1024), 6, cell=32).cuda()
    set_compute_dtype(torch.bfloat16)
    try:
        out = net(x)
        assert out.shape == (2, 6, 1024, 1024)
        loss = OhemCrossEntropyLoss(0, 0.7, 0.1)(out, lab)
        loss.backward()
    finally:
        set_compute_dtype(None)
    assert torch.isfinite(loss)
    dead = [n for n, p in net.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all() or float(p.grad.abs().max()) == 0.0]
    # biases that sit directly before a training-mode BatchNorm have an analytically zero gradient
    dead = [n for n in dead if not (n.endswith('.bias') and ('lawin' in n or 'conv' in n))]
    assert not dead, dead[:8]
