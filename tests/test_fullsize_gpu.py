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


def _dead(net):
    """parameters the step did not reach: no gradient, a non-finite one, or an all-zero one.  An all-zero gradient is legitimate for
    the parameters of ONE residual branch of an encoder block when DropPath dropped that branch for every image of the (2-image)
    batch; a few such branches are tolerated, anywhere else a zero gradient counts as dead."""
    bad, zero = [], []
    for n, p in net.named_parameters():
        if p.grad is None or not torch.isfinite(p.grad).all():
            bad.append(n)
        elif float(p.grad.abs().max()) == 0.0:
            zero.append(n)
    dropped = set()
    for n in zero:
        parts = n.split('.')
        if 'encoder' in parts and any(q.startswith('block') for q in parts):
            i = next(k for k, q in enumerate(parts) if q.startswith('block'))
            branch = 'attn' if parts[i + 2] in ('attn', 'norm1') else 'mlp'
            dropped.add('.'.join(parts[:i + 2] + [branch]))
        else:
            bad.append(n)
    if len(dropped) > 4:
        bad += sorted(dropped)
    return bad


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
    dead = _dead(net)
    # biases that sit directly before a training-mode BatchNorm have an analytically zero gradient
    dead = [n for n in dead if not (n.endswith('.bias') and ('lawin' in n or 'conv' in n))]
    assert not dead, dead[:8]


# ---------------------------------------------------------------------------- configs[2]: DNet(ConvNeXt-T + Decoder(embed 256)), 1024^2, bf16
def test_full_size_dnet_t_training_step_and_batch_independence():
    """ConvNeXt-T (dims up to 768, depth 9: never built by the small-fixture tests) + Decoder + terminator at 2 x 1024^2 in bf16:
    the probabilities of a training step sum to one at every pixel, the Dice step reaches every parameter, and eval-mode
    probabilities of an image do not depend on its batch neighbours."""
    from semseg_amd.decoder import DNet
    from semseg_amd.loss import DiceLoss
    from semseg_amd.runtime import set_compute_dtype
    from semseg_amd.detinit import det_fill_, det_input, det_labels
    net = det_fill_(DNet(3, 6, embed_dim=256, stem_code='T')).cuda().train()
    x = det_input('full.dnet.x', (2, 3, 1024, 1024), 0, 1).cuda()
    lab = det_labels('full.dnet.lab', (2, 1024, 1024), 6, cell=32).cuda()
    set_compute_dtype(torch.bfloat16)
    try:
        probs = net(x)
        assert probs.shape == (2, 6, 1024, 1024) and probs.dtype == torch.float32          # softmax output is fp32, as under autocast
        assert float((probs.sum(1) - 1).abs().max()) < 1e-5 and float(probs.min()) >= 0
        loss = DiceLoss(0.5, False, 6)(probs, lab)
        loss.backward()
        assert torch.isfinite(loss) and 0 < float(loss) < 1
        # a bias in front of a training-mode BatchNorm has an analytically zero gradient
        dead = [n for n in _dead(net) if not (n.endswith('.bias') and ('lawin' in n or 'conv' in n))]
        assert not dead, dead[:8]
        net.eval()
        with torch.no_grad():
            both = net(x).float()
            one = [net(x[i:i + 1]).float() for i in range(2)]
    finally:
        set_compute_dtype(None)
    for i in range(2):
        assert float((both[i] - one[i][0]).abs().max()) <= 2.0 ** -7, i          # probabilities: absolute, one bf16 rounding step of a split-K plan
    assert float((both[0] - both[1]).abs().max()) > 1e-3


# ---------------------------------------------------------------------------- configs[4]: MiT-B5 + Lawin(512), 2048^2, bf16, OHEM-CE + Dice
def test_full_size_mit_b5_2048_ce_dice_step_and_confusion_matrix():
    """MiT-B5 (depths 3/6/40/3) at 2 x 2048^2: N = 262144 queries, Nk = 4096 keys -- the regime of the attention backward that no
    small fixture reaches.  One OHEM-CE + Dice training step reaches every parameter with finite gradients; the confusion matrix of
    the step's logits adds up to the pixel count with the label histogram as its row sums; eval logits are batch-independent."""
    from semseg_amd.model import SegFormerLawin
    from semseg_amd.loss import OhemCEDiceLoss
    from semseg_amd.metrics import SegmentationMetrics
    from semseg_amd.runtime import set_compute_dtype
    from semseg_amd.detinit import det_fill_, det_input, det_labels
    net = SegFormerLawin('B5', 512, 6)
    det_fill_(net.encoder), det_fill_(net.decoder)
    net = net.cuda().train()
    assert [len(getattr(net.encoder, f'block{i}')) for i in (1, 2, 3, 4)] == [3, 6, 40, 3]
    x = det_input('full.b5.x', (2, 3, 2048, 2048), 0, 1).cuda()
    lab = det_labels('full.b5.lab', (2, 2048, 2048), 6, cell=64).cuda()
    set_compute_dtype(torch.bfloat16)
    try:
        out = net(x)
        assert out.shape == (2, 6, 2048, 2048)
        loss = OhemCEDiceLoss(6, 0, 0.7, 0.1)(out, lab)
        loss.backward()
        assert torch.isfinite(loss)
        dead = [n for n in _dead(net) if not (n.endswith('.bias') and ('lawin' in n or 'conv' in n))]
        assert not dead, dead[:8]
        met = SegmentationMetrics(6, ignore_index=0).cuda()
        met.update(out.detach(), lab)
        cm = met.confmat
        assert int(cm.sum()) == 2 * 2048 * 2048
        assert torch.equal(cm.sum(1), torch.bincount(lab.reshape(-1), minlength=6))
        assert torch.equal(cm.sum(0), torch.bincount(out.detach().float().argmax(1).reshape(-1), minlength=6))
        del out, loss
        net.eval()
        with torch.no_grad():
            both = net(x).float()
            one0 = net(x[:1]).float()
    finally:
        set_compute_dtype(None)
    scale = float(both.abs().max())
    assert torch.isfinite(both).all()
    assert float((both[0] - one0[0]).abs().max()) <= 2.0 ** -7 * scale
    assert float((both[0] - both[1]).abs().max()) > 1e-3 * scale


def test_full_size_topformer_b_refiner_training_step_properties():
    """The model train.py:45-52 builds (TopFormer-B + RefinerHead) at train.py's tile size, 1024 x 1024, batch 4, bf16: one training step
    (OHEM-CE) leaves a finite, non-zero gradient in every parameter and advances every BatchNorm's running statistics exactly once; a second
    step from the same state is bitwise identical (fixed-order reductions everywhere); and in fp32 the gradient is linear in the loss scale
    (backward(2 L) == 2 backward(L) up to fp32 rounding) -- a property of the whole backward chain that needs no oracle at this size."""
    from semseg_amd.topformer import TopFormerModel
    from semseg_amd.loss import OhemCrossEntropyLoss
    from semseg_amd.runtime import set_compute_dtype
    from semseg_amd.detinit import det_fill_, det_input, det_labels
    net = det_fill_(TopFormerModel(6, 'B', head_cfg=frozenset({'alias': 'refiner'}.items()))).cuda().train()
    x = det_input('full.tx', (4, 3, 1024, 1024), 0, 1).cuda()
    lab = det_labels('full.tl', (4, 1024, 1024), 6, cell=32).cuda()
    crit = OhemCrossEntropyLoss(0, 0.7, 0.1)
    state = {k: v.clone() for k, v in net.state_dict().items()}

    def step(dtype, scale=1.0):
        net.load_state_dict(state)
        net.zero_grad()
        set_compute_dtype(dtype)
        try:
            out = net(x)
            loss = crit(out, lab) * scale
            loss.backward()
        finally:
            set_compute_dtype(None)
        return out.detach().float(), float(loss.detach()), {n: p.grad.clone() for n, p in net.named_parameters()}

    out, loss, g1 = step(torch.bfloat16)
    assert out.shape == (4, 6, 1024, 1024) and bool(torch.isfinite(out).all()) and 1.0 < loss < 3.5
    assert _dead(net) == []
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            assert int(m.num_batches_tracked) == 1
    out2, loss2, g2 = step(torch.bfloat16)
    assert torch.equal(out, out2) and loss == loss2 and all(torch.equal(g1[n], g2[n]) for n in g1)
    _, la, ga = step(torch.float32)
    _, lb, gb = step(torch.float32, 2.0)
    assert abs(lb - 2 * la) < 1e-5 * abs(la)
    cat = lambda d: torch.cat([d[n].flatten() for n in sorted(d)])
    assert float((cat(gb) - 2 * cat(ga)).norm() / cat(ga).norm()) < 1e-5
