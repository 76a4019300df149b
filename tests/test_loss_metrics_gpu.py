"""OHEM-CE / Dice losses and the confusion-matrix metrics on the GPU vs the reference golden vectors and the oracle."""
import os
import numpy as np
import pytest
import torch

from semseg_amd.detinit import det_input, det_labels

pytestmark = pytest.mark.gpu


from _parity import rel, rel_l2  # noqa: E402,F401  (recording error measures)


def test_ohem_ce_matches_reference_golden(golden_dir):
    from semseg_amd.loss import OhemCrossEntropyLoss
    g = np.load(os.path.join(golden_dir, 'losses.npz'))
    logits = torch.from_numpy(g['logits'])
    lab, lab_ign = torch.from_numpy(g['lab']), torch.from_numpy(g['lab_many_ignored'])
    cases = {'plain': (lab, 0.7, 0.1, 0), 'ignored': (lab_ign, 0.7, 0.1, 0), 'nokeep': (lab, 0.7, 0.0, 0),
             'half': (lab, 0.3, 0.5, 0), 'noignore': (lab, 0.7, 0.25, 255)}
    for name, (tgt, thr, mk, ign) in cases.items():
        crit = OhemCrossEntropyLoss(ign, thr, mk)
        for call in range(2):                       # the second call sees the ratcheted threshold of the first
            lg = (logits * (1.0 + call)).cuda().requires_grad_(True)
            loss = crit(lg, tgt.cuda())
            loss.backward()
            assert rel(loss, g[f'ohem.{name}.{call}.loss']) < 1e-5, (name, call)
            assert rel(lg.grad, g[f'ohem.{name}.{call}.grad']) < 1e-4, (name, call)
            assert np.float32(crit.threshold) == g[f'ohem.{name}.{call}.thr'], (name, call)


@pytest.mark.parametrize('C', [7, 6, 8, 2])
def test_ohem_ce_strided_channels_last_and_bf16_vs_oracle(C):
    """NHWC-strided logits (what the networks return): the channels-last kernels, scalar (odd C) and 4-byte-pair (even C, bf16) loads."""
    from oracle.losses import OhemCE
    from semseg_amd.loss import OhemCrossEntropyLoss
    logits = det_input('l2.logits', (3, C, 40, 24), -4, 4)
    lab = det_labels('l2.lab', (3, 40, 24), C, cell=4)
    for dtype, tol in ((torch.float32, 1e-5), (torch.bfloat16, 1e-2)):
        lq = logits.to(dtype).float()
        ref_in = lq.clone().requires_grad_(True)
        ref = OhemCE(1, 0.6, 0.3)(ref_in, lab)
        ref.backward()
        cl = lq.to(dtype).cuda().permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).requires_grad_(True)   # NHWC strides
        loss = OhemCrossEntropyLoss(1, 0.6, 0.3)(cl, lab.cuda())
        loss.backward()
        assert rel(loss, ref) < tol and rel(cl.grad, ref_in.grad) < max(tol, 1e-4) * 4


def test_ohem_full_size_properties():
    """BASELINE-size check through size-independent properties: 4x1024x1024 pixels, kept count and loss vs a torch
    restatement of the selection on the GPU tensor itself (sort-based)."""
    from semseg_amd.loss import OhemCrossEntropyLoss
    torch.manual_seed(0)
    B, C, S = 4, 6, 1024
    logits = torch.randn(B, C, S, S, device='cuda') * 2
    lab = torch.randint(0, C, (B, S // 32, S // 32), device='cuda').repeat_interleave(32, 1).repeat_interleave(32, 2)
    crit = OhemCrossEntropyLoss(0, 0.7, 0.1)
    loss = crit(logits, lab)
    prob = logits.softmax(1).gather(1, lab[:, None]).reshape(-1)
    valid = lab.reshape(-1) != 0
    vp = prob[valid]
    k = int(0.1 * vp.numel())
    idx = vp.argsort(stable=True)[k - 1]
    thr = max(0.7, float(prob[idx]))
    keep = valid & (prob < thr)
    nll = torch.nn.functional.cross_entropy(logits, lab, reduction='none').reshape(-1)
    assert abs(float(loss) - float(nll[keep].double().mean())) < 1e-5 * float(loss)
    assert abs(crit.threshold - thr) < 1e-7


def test_dice_losses_match_reference_golden(golden_dir):
    from semseg_amd.loss import DiceLoss, dice_loss, class_weights
    g = np.load(os.path.join(golden_dir, 'losses.npz'))
    probs = torch.from_numpy(g['logits']).softmax(1)
    lab = torch.from_numpy(g['lab']).cuda()
    for weighted in (0, 1):
        for delta in (0.5, 0.3):
            p = probs.cuda().requires_grad_(True)
            loss = DiceLoss(delta, bool(weighted), 6)(p, lab)
            loss.backward()
            assert rel(loss, g[f'dice.{weighted}.{delta}.loss']) < 1e-5
            assert rel(p.grad, g[f'dice.{weighted}.{delta}.grad']) < 1e-4
    for smooth in (0.0, 1.0):
        p = probs.cuda().requires_grad_(True)
        loss = dice_loss(p, lab, smooth)
        loss.backward()
        assert rel(loss, g[f'dicef.{smooth}.loss']) < 1e-5
        assert rel(p.grad, g[f'dicef.{smooth}.grad']) < 1e-4
    w = class_weights(torch.from_numpy(g['lab_many_ignored']).cuda(), torch.arange(8).cuda())
    assert rel(w, g['class_weights8']) < 1e-6
    # any class ids in any order (the reference compares labels with the ids it is given, helper/weights.py:22)
    from oracle import losses as olosses
    ids = torch.tensor([5, 2, 7, 0, 3])
    w2 = class_weights(torch.from_numpy(g['lab']).cuda(), ids.cuda())
    assert rel(w2, olosses.class_weights(torch.from_numpy(g['lab']), ids)) < 1e-6
    with pytest.raises(ValueError):
        class_weights(lab, torch.tensor([0, 5000]).cuda())


def test_label_hist_exact_full_size():
    from semseg_amd import ops
    torch.manual_seed(3)
    lab = torch.randint(-2, 9, (16, 1024, 1024), device='cuda')            # values outside 0..5 are not counted
    h = ops.label_hist(lab, 6)
    want = torch.bincount(lab[(lab >= 0) & (lab < 6)].reshape(-1), minlength=6)
    assert h.dtype == torch.int64 and torch.equal(h, want)
    assert torch.equal(ops.label_hist(lab[:1, :3, :5].contiguous() * 0 + 4095, 4096)[4095:], torch.tensor([15], device='cuda'))


def test_confusion_matrix_bit_exact_and_scores():
    from oracle import metrics as om
    from semseg_amd.metrics import SegmentationMetrics
    logits = det_input('m.logits', (2, 6, 64, 48), -3, 3)
    logits[:, 3] = logits[:, 1]                       # ties: the first maximum must win
    lab = det_labels('m.lab', (2, 64, 48), 6, cell=4)
    m = SegmentationMetrics(6, ignore_index=0, prefix='Training', delimiter='-').cuda()
    for dtype in (torch.float32, torch.bfloat16):
        m.reset()
        lg = logits.to(dtype)
        m.update(preds=lg.cuda(), target=lab.cuda())
        m.update(preds=lg.cuda().permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2), target=lab.cuda())   # channels-last view
        ref = om.confusion_matrix(lg.float(), lab, 6) * 2
        assert torch.equal(m.confmat.cpu(), ref)                       # bit-exact integer counts
    out = m.compute()
    d = om.derive(ref.numpy())
    for k in ('Precision', 'Recall', 'IoU', 'F1', 'Accuracy'):
        assert np.allclose(out[f'Training-{k}'].cpu().numpy(), d[k], atol=1e-6, equal_nan=True), k
    assert float(out['Training-mIoU']) == pytest.approx(float(d['mIoU']), abs=1e-6)
    assert float(out['Training-Global Accuracy']) == pytest.approx(float(d['Global Accuracy']), abs=1e-6)
    lg_ref = om.loggable(ref.numpy(), ignore_index=0)
    lg_out = m.loggable_dict()
    for k, v in lg_ref.items():
        assert float(lg_out[f'Training-{k}']) == pytest.approx(float(v), rel=1e-5), k
    rep = m.tabular_report()
    assert rep['confusion_matrix'].shape == (6, 6) and rep['class_report'].shape == (5, 5)


def test_confusion_matrix_from_class_indices_equals_scores_path():
    """helper/metrics.py:90-95: torchmetrics' ConfusionMatrix takes class-index predictions as well as scores; update() with
    argmax(preds, 1) must add exactly what update() with the scores adds (bit-exact counts), out-of-range pairs dropped."""
    from oracle import metrics as om
    from semseg_amd.metrics import SegmentationMetrics
    logits = det_input('mi.logits', (3, 6, 40, 56), -3, 3)
    lab = det_labels('mi.lab', (3, 40, 56), 6, cell=4)
    lab[0, :3] = 255                                   # a target outside 0..5 is not counted by either path
    ref = om.confusion_matrix(logits, lab.clamp(max=5), 6) - om.confusion_matrix(logits[:1, :, :3], lab[:1, :3].clamp(max=5), 6)
    m = SegmentationMetrics(6).cuda()
    m.update(preds=logits.argmax(1).cuda(), target=lab.cuda())
    assert torch.equal(m.confmat.cpu(), ref)
    m.update(preds=logits.argmax(1).to(torch.uint8).cuda(), target=lab.cuda())      # an integer class map of another width
    assert torch.equal(m.confmat.cpu(), 2 * ref)
    with pytest.raises(ValueError):
        m.update(preds=logits[:, 0].cuda(), target=lab.cuda())                      # float [B, H, W] is neither scores nor indices


def test_confusion_matrix_full_size_checksum():
    """16 x 1024 x 1024 pixels: total count, row sums (label histogram) and column sums (prediction histogram)."""
    from semseg_amd.metrics import SegmentationMetrics
    torch.manual_seed(1)
    B, C, S = 16, 6, 1024
    logits = torch.randn(B, C, S, S, device='cuda', dtype=torch.bfloat16)
    lab = torch.randint(0, C, (B, S, S), device='cuda')
    m = SegmentationMetrics(C).cuda()
    m.update(logits, lab)
    cm = m.confmat
    assert int(cm.sum()) == B * S * S
    assert torch.equal(cm.sum(1), torch.bincount(lab.reshape(-1), minlength=C))
    assert torch.equal(cm.sum(0), torch.bincount(logits.float().argmax(1).reshape(-1), minlength=C))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', [(2, 6, 33, 47), (1, 19, 8, 8), (3, 2, 5, 64)])
def test_softmax2d_fwd_bwd_vs_torch(dtype, shape):
    from semseg_amd.loss import Softmax2d
    x = det_input('sm.x', shape, -4, 4)
    w = det_input('sm.w', shape, -1, 1)
    xr = x.to(dtype).float().clone().requires_grad_(True)
    ref = torch.nn.Softmax2d()(xr)
    (ref * w).sum().backward()
    for layout in ('nchw', 'nhwc'):
        xg = x.detach().to(dtype).cuda()
        if layout == 'nhwc':
            xg = xg.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        xg.requires_grad_(True)
        y = Softmax2d()(xg)
        (y.float() * w.cuda()).sum().backward()
        tol = 1e-6 if dtype == torch.float32 else 1e-2
        assert float((y.detach().float().cpu() - ref.detach()).abs().max()) < tol
        assert float((xg.grad.float().cpu() - xr.grad).abs().max()) < tol * max(1.0, float(xr.grad.abs().max()))


def test_ohem_ce_plus_dice_vs_oracle():
    """BASELINE configs[4] objective: OHEM-CE(logits) + Dice(softmax(logits)), loss and logit gradient vs the CPU oracle (fp32 1e-4)."""
    from oracle.losses import OhemCE, tversky_dice
    from semseg_amd.loss import OhemCEDiceLoss
    logits = det_input('cd.logits', (2, 6, 48, 40), -3, 3)
    lab = det_labels('cd.lab', (2, 48, 40), 6, cell=4)
    lo = logits.clone().requires_grad_(True)
    want = OhemCE(0, 0.7, 0.1)(lo, lab) + tversky_dice(torch.softmax(lo, dim=1), lab, 6, 0.5, False)
    want.backward()
    lg = logits.cuda().permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).requires_grad_(True)
    got = OhemCEDiceLoss(6, 0, 0.7, 0.1)(lg, lab.cuda())
    got.backward()
    assert rel(got, want) < 1e-5
    assert rel(lg.grad, lo.grad) < 1e-4
