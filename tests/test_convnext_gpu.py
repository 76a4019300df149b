"""ConvNeXt encoder, Decoder and DNet on the GPU vs the reference golden vectors (tests/golden/make_golden.py) and the CPU oracle."""
import os
import numpy as np
import pytest
import torch

from semseg_amd.detinit import det_fill_, det_input, det_labels, det_state_dict
from semseg_amd import shapes

pytestmark = pytest.mark.gpu


from _parity import rel, rel_l2  # noqa: E402,F401  (recording error measures)



def test_state_dict_layout_matches_reference_manifest():
    from semseg_amd.convnext import ConvNeXt
    from semseg_amd.decoder import DNet
    assert {k: tuple(v.shape) for k, v in ConvNeXt('X').state_dict().items()} == shapes.convnext_shapes('X', 3)
    assert {k: tuple(v.shape) for k, v in DNet(3, 6, 128, 'X').state_dict().items()} == shapes.dnet_shapes(3, 6, 128, 'X')


def test_convnext_x_fp32_matches_reference_golden(golden_dir):
    from semseg_amd.convnext import ConvNeXt
    g = np.load(os.path.join(golden_dir, 'convnext_x_64.npz'))
    enc = det_fill_(ConvNeXt('X')).cuda().eval()
    outs = enc(det_input('cnx.x', (2, 3, 64, 64)).cuda())
    projs = [det_input(f'cnx.proj{i}', o.shape, -1, 1).cuda() for i, o in enumerate(outs)]
    sum((o * p).sum() for o, p in zip(outs, projs)).backward()
    for i in range(4):
        assert outs[i].shape == g[f'out{i}'].shape and rel(outs[i], g[f'out{i}']) < 1e-4, i
    params = dict(enc.named_parameters())
    for k in g.files:
        if k.startswith('grad.'):
            assert rel(params[k[5:]].grad, g[k]) < 1e-4, k


def test_convnext_x_bf16_within_budget(golden_dir):
    from semseg_amd.convnext import ConvNeXt
    from semseg_amd.runtime import set_compute_dtype
    g = np.load(os.path.join(golden_dir, 'convnext_x_64.npz'))
    enc = det_fill_(ConvNeXt('X')).cuda().eval()
    set_compute_dtype(torch.bfloat16)
    try:
        outs = enc(det_input('cnx.x', (2, 3, 64, 64)).cuda())
        projs = [det_input(f'cnx.proj{i}', o.shape, -1, 1).cuda() for i, o in enumerate(outs)]
        sum((o.float() * p).sum() for o, p in zip(outs, projs)).backward()
    finally:
        set_compute_dtype(None)
    for i in range(4):
        assert outs[i].dtype == torch.bfloat16 and rel_l2(outs[i], g[f'out{i}']) < 2e-2, i
    params = dict(enc.named_parameters())
    for k in ('grad.stages.1.0.pw_conv1.weight', 'grad.stages.3.1.dw_conv.weight', 'grad.downsample_layers.2.1.weight'):
        assert rel_l2(params[k[5:]].grad, g[k]) < max(2e-2, 1.5 * rel_l2(g['bf16.' + k], g[k])), k     # see test_mit_gpu.py


def test_dnet_training_step_fp32_matches_reference_golden(golden_dir):
    from semseg_amd.decoder import DNet
    from semseg_amd.loss import DiceLoss
    g = np.load(os.path.join(golden_dir, 'dnet_x_128.npz'))
    net = det_fill_(DNet(3, 6, embed_dim=128, stem_code='X')).cuda().train()
    net.decoder.decoder.dropout.p = 0.0
    x = det_input('dnet.x', (2, 3, 128, 128)).cuda()
    lab = det_labels('dnet.lab', (2, 128, 128), 6, cell=8).cuda()
    probs = net(x)
    assert probs.shape == (2, 6, 128, 128) and rel(probs, g['probs']) < 1e-4
    assert float((probs.float().sum(1) - 1).abs().max()) < 1e-5
    loss = DiceLoss(0.5, False, 6)(probs, lab)
    loss.backward()
    assert rel(loss, g['loss']) < 1e-4
    params = dict(net.named_parameters())
    for k in g.files:
        if k.startswith('grad.'):
            assert rel(params[k[5:]].grad, g[k]) < 1e-4, k
    sd = net.state_dict()
    for k in g.files:
        if k.startswith('after.'):
            assert rel(sd[k[6:]].float(), g[k].astype(np.float32)) < 1e-4, k


def test_dnet_training_step_bf16_within_reference_bf16_budget(golden_dir):
    """The bf16 instantiations of the decoder / terminator kernels (what configs[2] runs on): probabilities within 2e-2 of the fp32
    reference; every listed gradient within max(2e-2, 1.5 x the deviation of the reference's OWN bf16-autocast run from its fp32
    run) -- the fixture's bf16.* vectors: the chain of training-mode BatchNorms makes that deviation 2e-2 .. 2.5e-1 here."""
    from semseg_amd.decoder import DNet
    from semseg_amd.loss import DiceLoss
    from semseg_amd.runtime import set_compute_dtype
    g = np.load(os.path.join(golden_dir, 'dnet_x_128.npz'))
    net = det_fill_(DNet(3, 6, embed_dim=128, stem_code='X')).cuda().train()
    net.decoder.decoder.dropout.p = 0.0
    x = det_input('dnet.x', (2, 3, 128, 128)).cuda()
    lab = det_labels('dnet.lab', (2, 128, 128), 6, cell=8).cuda()
    set_compute_dtype(torch.bfloat16)
    try:
        probs = net(x)
        loss = DiceLoss(0.5, False, 6)(probs, lab)
        loss.backward()
    finally:
        set_compute_dtype(None)
    assert probs.dtype == torch.float32 and probs.shape == (2, 6, 128, 128)      # softmax output is fp32, as under torch.autocast
    assert rel_l2(probs, g['probs']) < 2e-2
    assert float((probs.sum(1) - 1).abs().max()) < 1e-5
    assert rel(loss, g['loss']) < 2e-2
    params = dict(net.named_parameters())
    for k in g.files:
        if k.startswith('grad.'):
            ref_dev = float(np.linalg.norm(g['bf16.' + k] - g[k]) / np.linalg.norm(g[k]))
            assert params[k[5:]].grad.dtype == torch.float32
            assert rel_l2(params[k[5:]].grad, g[k]) < max(2e-2, 1.5 * ref_dev), (k, ref_dev)


def test_decoder_eval_nonsquare_vs_oracle():
    """Decoder alone (eval-mode BatchNorms, running statistics), non-square 128x192 tile: output (H-1) x (W-1)."""
    from oracle import convnext as oconvnext
    from semseg_amd.decoder import Decoder
    ch = [32, 64, 128, 256]
    dec = det_fill_(Decoder(ch, 12, 128)).cuda().eval()
    feats_cpu = [det_input(f'dec.f{i}', (2, c, 32 >> i, 48 >> i), -1, 1) for i, c in enumerate(ch)]
    feats = [f.cuda().requires_grad_(True) for f in feats_cpu]
    out = dec(feats)
    assert out.shape == (2, 12, 127, 191)
    proj = det_input('dec.proj', out.shape, -1, 1)
    (out * proj.cuda()).sum().backward()
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v.clone())
          for k, v in det_state_dict(shapes.decoder_shapes(ch, 12, 128)).items()}
    fr = [f.clone().requires_grad_(True) for f in feats_cpu]
    oout = oconvnext.decoder_forward(sd, fr, training=False)
    (oout * proj).sum().backward()
    assert rel(out, oout) < 1e-4
    for i in range(4):
        assert rel(feats[i].grad, fr[i].grad) < 1e-4, i
    for k in ('deconvolution.net.0.weight', 'deconvolution.net.0.bias', 'deconvolution.net.1.weight', 'feature_reducer.net.p_conv.conv.weight'):
        assert rel(dict(dec.named_parameters())[k].grad, sd[k].grad) < 1e-4, k
