"""The CPU oracle (oracle/) must reproduce the golden vectors that
tests/golden/make_golden.py captured from the reference itself."""
import os
import numpy as np
import torch
import torch.nn.functional as F
import pytest

from semseg_amd.detinit import det_state_dict, det_input, det_labels
from semseg_amd import shapes
from oracle import mit as omit, lawin as olawin, losses as olosses, metrics as ometrics


def rel(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def grad_sd(sd):
    return {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v.clone())
            for k, v in sd.items()}


def test_mit_b0(golden_dir):
    g = np.load(os.path.join(golden_dir, 'mit_b0_64.npz'))
    sd = grad_sd(det_state_dict(shapes.mit_shapes('B0', 3)))
    x = torch.from_numpy(g['x'])
    assert rel(x, det_input('mit.x', (2, 3, 64, 64))) == 0
    outs = omit.mit_forward(sd, x, 'B0')
    projs = [det_input(f'mit.proj{i}', o.shape, -1, 1) for i, o in enumerate(outs)]
    sum((o * p).sum() for o, p in zip(outs, projs)).backward()
    for i in range(4):
        assert rel(outs[i].detach(), g[f'out{i}']) < 2e-5
    for k in g.files:
        if k.startswith('grad.'):
            assert rel(sd[k[5:]].grad, g[k]) < 5e-5, k


@pytest.mark.parametrize('training', [False, True])
def test_lawin_head(golden_dir, training):
    tag = 'train' if training else 'eval'
    g = np.load(os.path.join(golden_dir, f'lawin_e128_{tag}.npz'))
    chans = [32, 64, 160, 256]
    sd = grad_sd(det_state_dict(shapes.lawin_shapes(chans, 128, 6)))
    feats = [det_input(f'head.f{i}', (2, c, 32 >> i, 32 >> i), -1, 1).requires_grad_(True)
             for i, c in enumerate(chans)]
    out = olawin.lawin_head(sd, feats, training=training)
    (out * det_input('head.proj', out.shape, -1, 1)).sum().backward()
    assert rel(out.detach(), g['out']) < 2e-5
    for i in range(4):
        assert rel(feats[i].grad, g[f'dfeat{i}']) < 5e-5
    for k in g.files:
        if k.startswith('grad.'):
            assert rel(sd[k[5:]].grad, g[k]) < 5e-5, k
        if k.startswith('after.'):
            assert rel(sd[k[6:]].float(), g[k].astype(np.float32)) < 1e-5, k


def test_lawin_head_e256(golden_dir):
    g = np.load(os.path.join(golden_dir, 'lawin_e256_eval.npz'))
    chans = [32, 64, 160, 256]
    sd = grad_sd(det_state_dict(shapes.lawin_shapes(chans, 256, 6)))
    feats = [det_input(f'head256.f{i}', (2, c, 32 >> i, 32 >> i), -1, 1).requires_grad_(True) for i, c in enumerate(chans)]
    out = olawin.lawin_head(sd, feats, training=False)
    (out * det_input('head256.proj', out.shape, -1, 1)).sum().backward()
    assert rel(out.detach(), g['out']) < 2e-5
    for i in range(4):
        assert rel(feats[i].grad, g[f'dfeat{i}']) < 5e-5


def test_segmentor_and_ohem(golden_dir):
    g = np.load(os.path.join(golden_dir, 'segmentor_b0_e128_128.npz'))
    sd_e = det_state_dict(shapes.mit_shapes('B0', 3))
    sd_h = det_state_dict(shapes.lawin_shapes([32, 64, 160, 256], 128, 6))
    x = det_input('seg.x', (2, 3, 128, 128))
    out = olawin.segmentor(sd_e, sd_h, x, 'B0')
    assert rel(out, g['logits']) < 2e-5
    lab = det_labels('seg.lab', (2, 128, 128), 6, cell=8)
    loss = olosses.OhemCE(0, 0.7, 0.1)(out, lab)
    assert rel(loss, g['loss']) < 1e-5
    # the reference's own bf16-autocast run sits inside the 2e-2 budget of the north star (rel-L2)
    assert float(np.linalg.norm(g['logits_bf16'] - g['logits']) / np.linalg.norm(g['logits'])) < 2e-2


def test_losses(golden_dir):
    g = np.load(os.path.join(golden_dir, 'losses.npz'))
    logits = torch.from_numpy(g['logits'])
    lab, lab_ign = torch.from_numpy(g['lab']), torch.from_numpy(g['lab_many_ignored'])
    cases = {'plain': (lab, 0.7, 0.1, 0), 'ignored': (lab_ign, 0.7, 0.1, 0), 'nokeep': (lab, 0.7, 0.0, 0),
             'half': (lab, 0.3, 0.5, 0), 'noignore': (lab, 0.7, 0.25, 255)}
    for name, (tgt, thr, mk, ign) in cases.items():
        crit = olosses.OhemCE(ign, thr, mk)
        for call in range(2):
            lg = (logits * (1.0 + call)).clone().requires_grad_(True)
            loss = crit(lg, tgt)
            loss.backward()
            assert rel(loss.detach(), g[f'ohem.{name}.{call}.loss']) < 1e-6
            assert rel(lg.grad, g[f'ohem.{name}.{call}.grad']) < 1e-5
            assert np.float32(float(crit.threshold)) == g[f'ohem.{name}.{call}.thr']
    probs = logits.softmax(1)
    for weighted in (0, 1):
        for delta in (0.5, 0.3):
            p = probs.clone().requires_grad_(True)
            loss = olosses.tversky_dice(p, lab, 6, delta, bool(weighted))
            loss.backward()
            assert rel(loss.detach(), g[f'dice.{weighted}.{delta}.loss']) < 1e-6
            assert rel(p.grad, g[f'dice.{weighted}.{delta}.grad']) < 1e-5
    for smooth in (0.0, 1.0):
        p = probs.clone().requires_grad_(True)
        loss = olosses.dice_functional(p, lab, smooth)
        loss.backward()
        assert rel(loss.detach(), g[f'dicef.{smooth}.loss']) < 1e-6
        assert rel(p.grad, g[f'dicef.{smooth}.grad']) < 1e-5
    assert rel(olosses.class_weights(lab_ign, torch.arange(8)), g['class_weights8']) < 1e-6


def test_metrics_against_sklearn():
    from sklearn.metrics import confusion_matrix, jaccard_score, f1_score, precision_score, recall_score
    logits = det_input('m.logits', (2, 6, 32, 32), -3, 3)
    logits[:, 3] = logits[:, 1]                # force ties -> first max index must win
    lab = det_labels('m.lab', (2, 32, 32), 6, cell=4)
    cm = ometrics.confusion_matrix(logits, lab, 6)
    pred = logits.argmax(1).reshape(-1).numpy()
    y = lab.reshape(-1).numpy()
    assert (pred != 3).all() or (logits[:, 3] > logits[:, 1]).any()
    assert np.array_equal(cm.numpy(), confusion_matrix(y, pred, labels=list(range(6))))
    d = ometrics.derive(cm.numpy())
    lbl = list(range(6))
    assert np.allclose(d['IoU'], jaccard_score(y, pred, labels=lbl, average=None, zero_division=1.0), atol=1e-6)
    present = ~np.isnan(d['F1'])
    assert np.allclose(d['F1'][present], f1_score(y, pred, labels=lbl, average=None, zero_division=0)[present], atol=1e-6)
    assert np.allclose(d['Precision'][present], precision_score(y, pred, labels=lbl, average=None, zero_division=0)[present], atol=1e-6)
    assert np.allclose(d['Recall'][present], recall_score(y, pred, labels=lbl, average=None, zero_division=0)[present], atol=1e-6)
    assert abs(d['Global Accuracy'] - (pred == y).mean()) < 1e-6
    lg = ometrics.loggable(cm.numpy(), ignore_index=0)
    assert abs(lg['Micro_Accuracy'] - np.diag(cm.numpy()).mean()) < 1e-3


def test_convnext_x(golden_dir):
    from oracle import convnext as oconvnext
    g = np.load(os.path.join(golden_dir, 'convnext_x_64.npz'))
    sd = grad_sd(det_state_dict(shapes.convnext_shapes('X', 3)))
    x = det_input('cnx.x', (2, 3, 64, 64))
    outs = oconvnext.convnext_forward(sd, x, 'X')
    projs = [det_input(f'cnx.proj{i}', o.shape, -1, 1) for i, o in enumerate(outs)]
    sum((o * p).sum() for o, p in zip(outs, projs)).backward()
    for i in range(4):
        assert rel(outs[i].detach(), g[f'out{i}']) < 2e-5
    for k in g.files:
        if k.startswith('grad.'):
            assert rel(sd[k[5:]].grad, g[k]) < 5e-5, k


def test_dnet_x_training_step(golden_dir):
    from oracle import convnext as oconvnext
    g = np.load(os.path.join(golden_dir, 'dnet_x_128.npz'))
    sd = grad_sd(det_state_dict(shapes.dnet_shapes(3, 6, 128, 'X')))
    x = det_input('dnet.x', (2, 3, 128, 128))
    lab = det_labels('dnet.lab', (2, 128, 128), 6, cell=8)
    probs = oconvnext.dnet_forward(sd, x, 'X', training=True)
    loss = olosses.tversky_dice(probs, lab, 6, 0.5, False)
    loss.backward()
    assert probs.shape == (2, 6, 128, 128) and rel(probs.detach(), g['probs']) < 2e-5 and rel(loss.detach(), g['loss']) < 1e-5
    for k in g.files:
        if k.startswith('grad.'):
            assert rel(sd[k[5:]].grad, g[k]) < 2e-4, k
        if k.startswith('after.'):
            assert rel(sd[k[6:]].float(), g[k].astype(np.float32)) < 1e-5, k
