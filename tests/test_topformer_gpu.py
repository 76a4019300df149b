"""TopFormer token pyramid + SimpleHead (inference) vs golden vectors captured from the reference (make_topformer_golden.py)."""
import json
import os
import numpy as np
import pytest
import torch

from semseg_amd.detinit import det_fill_, det_input


from _parity import rel, rel_l2  # noqa: E402,F401  (recording error measures)



VARIANTS = {'T_simple': ('T', 'sum', {'alias': 'simple'}), 'S_simple': ('S', 'sum', {'alias': 'simple'}), 'B_simple': ('B', 'sum', {'alias': 'simple'}),
            'B_refiner': ('B', 'sum', {'alias': 'refiner'}), 'T_simple_concat': ('T', 'concat', {'alias': 'simple', 'fusion': 'concat'})}


def make(tag):
    from semseg_amd.topformer import TopFormerModel
    alias, fusion, head = VARIANTS[tag]
    return TopFormerModel(6, alias, 3, 'dot_sum', fusion, frozenset(head.items()))


@pytest.mark.parametrize('tag', sorted(VARIANTS))
def test_state_dict_keys_match_reference_manifest(golden_dir, tag):
    """CPU-only: the parameter tree carries exactly the reference's state_dict keys and shapes (checkpoint contract)."""
    manifest = json.load(open(os.path.join(golden_dir, f'topformer_{tag}_manifest.json')))
    sd = make(tag).state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == manifest


def test_unsupported_variants_fail_loudly():
    from semseg_amd.topformer import TopFormerModel
    with pytest.raises(NotImplementedError):
        TopFormerModel(6, 'B', head_cfg=frozenset({'alias': 'nope'}.items()))
    with pytest.raises(ValueError):                 # the reference builds these and then fails in the first forward (channel mismatch)
        TopFormerModel(6, 'B', fusion='concat', head_cfg=frozenset({'alias': 'refiner'}.items()))
    with pytest.raises(ValueError):
        TopFormerModel(6, 'B', fusion='concat', head_cfg=frozenset({'alias': 'simple'}.items()))
    with pytest.raises(NotImplementedError):
        TopFormerModel(6, 'B', head_cfg=frozenset({'alias': 'refiner', 'act_cfg': frozenset({'alias': 'relu'}.items())}.items()))
    net = TopFormerModel(6, 'T').train()
    with pytest.raises((NotImplementedError, RuntimeError)):
        net(torch.zeros(1, 3, 64, 64))


@pytest.mark.gpu
@pytest.mark.parametrize('alias', ['T', 'S', 'B'])
def test_topformer_logits_match_reference_golden(golden_dir, alias):
    from semseg_amd.topformer import TopFormerModel
    from semseg_amd.runtime import set_compute_dtype
    g = np.load(os.path.join(golden_dir, f'topformer_{alias}_simple.npz'))
    net = det_fill_(TopFormerModel(6, alias)).cuda().eval()
    x = det_input('top.x', (2, 3, 128, 192)).cuda()
    out = net(x)
    assert out.shape == (2, 6, 128, 192) and net.out_channels == 6
    assert rel(out, g['logits']) < 1e-4
    feats = net.backbone(x)
    assert list(feats.keys()) == ['stage_1', 'stage_2', 'stage_3', 'stage_4']
    for k in ('stage_3', 'stage_4'):
        assert feats[k].shape == g['feat.' + k].shape and rel(feats[k], g['feat.' + k]) < 1e-4, k
    set_compute_dtype(torch.bfloat16)
    try:
        ob = net(x)
    finally:
        set_compute_dtype(None)
    assert ob.dtype == torch.bfloat16 and rel_l2(ob, g['logits']) < 2e-2
    # the module can be rebuilt in the same process (the reference cannot) and reloads a state_dict
    net2 = TopFormerModel(6, alias).cuda().eval()
    net2.load_state_dict(net.state_dict())
    assert torch.equal(net2(x), out)


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['B_refiner', 'T_simple_concat'])
def test_topformer_head_variants_match_reference_golden(golden_dir, tag):
    """RefinerHead (SimAM + grouped conv + two GroupNorms, modules/heads.py:173-269) and the concat-fusion SimpleHead."""
    from semseg_amd.runtime import set_compute_dtype
    g = np.load(os.path.join(golden_dir, f'topformer_{tag}.npz'))
    net = det_fill_(make(tag)).cuda().eval()
    x = det_input('top.x', (2, 3, 128, 192)).cuda()
    out = net(x)
    assert out.shape == (2, 6, 128, 192)
    assert rel(out, g['logits']) < 1e-4
    assert torch.equal(net(x), out)                                     # fixed-order reductions: bitwise reproducible
    set_compute_dtype(torch.bfloat16)
    try:
        ob = net(x)
    finally:
        set_compute_dtype(None)
    assert ob.dtype == torch.bfloat16 and rel_l2(ob, g['logits']) < 2e-2
    # images are independent in both heads (per-image statistics): a batch of one reproduces its slice of the batch of two
    assert rel(net(x[1:]), g['logits'][1:]) < 1e-4


@pytest.mark.gpu
def test_refiner_pieces_against_torch():
    """ss_image_moments / ss_simam_apply / ss_refiner_fold / ss_image_affine against fp32 torch restatements of modules/heads.py:27-39
    and nn.GroupNorm, on ragged sizes (19 classes -> 24 padded columns, 3 images, odd pixel counts)."""
    from semseg_amd import ops
    torch.manual_seed(0)
    b, n, C = 3, 35 * 9, 64
    x = (torch.randn(b * n, C) * 2 + torch.randn(C)).cuda()
    mom = ops.image_moments(x, b)
    xi = x.view(b, n, C)
    mu = xi.mean(1)
    assert rel(mom[:, 0], mu) < 1e-5 and rel(mom[:, 1], (xi - mu[:, None]).pow(2).sum(1)) < 1e-5
    k = torch.tensor([0.01], device='cuda')
    y = ops.simam_apply(x, mom, k, b)
    d2 = (xi - mu[:, None]).pow(2)
    want = xi * (d2 / (4 * (k + d2.sum(1, keepdim=True) / (n - 1))) + 0.5).sigmoid()
    assert rel(y.view(b, n, C), want) < 1e-5
    yb = ops.simam_apply(x.bfloat16(), ops.image_moments(x.bfloat16(), b), k, b)
    assert rel_l2(yb.view(b, n, C), want) < 1e-2
    # GroupNorm(4, 4*19) after a sigmoid, then a fixed linear map
    groups, cg, pitch = 4, 19, 24
    u = torch.zeros(b * n, groups * pitch, device='cuda')
    val = torch.randn(b * n, groups, cg, device='cuda')
    u.view(b * n, groups, pitch)[:, :, :cg] = val
    gn = torch.nn.GroupNorm(groups, groups * cg).cuda()
    with torch.no_grad():
        gn.weight.normal_(1, 0.2), gn.bias.normal_(0, 0.2)
    L = torch.randn(19, groups * cg, device='cuda')
    M, c = ops.refiner_fold(ops.image_moments(u, b, pre_act=ops.SIGMOID), n, groups, cg, pitch, gn.eps, gn.weight.detach(), gn.bias.detach(), L, 24)
    got = ops.image_affine(u, M, c, pre_act=ops.SIGMOID)
    s = val.view(b, n, groups * cg).sigmoid().permute(0, 2, 1).reshape(b, groups * cg, n, 1)
    want = torch.einsum('ro,bon->bnr', L, gn(s).detach().view(b, groups * cg, n))
    assert got.shape == (b * n, 24) and rel(got.view(b, n, 24)[..., :19], want) < 1e-4 and float(got[:, 19:].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('hw', [(64, 96), (37, 51)])
def test_stem_conv_direct_vs_torch_and_vs_im2col_gemm(dtype, hw):
    """The direct stem kernel (image -> 16 channels, k3 s2, AutoPadding (0, 1)) against torch's conv2d on the CPU in fp32 and
    against the im2col + GEMM path it replaces; fp32 1e-5, bf16 one rounding of the output (2^-8 relative)."""
    import torch.nn.functional as F
    from semseg_amd import ops
    H, W = hw
    g = torch.Generator().manual_seed(12)
    x = torch.rand((2, 3, H, W), generator=g)
    w = torch.randn((16, 3, 3, 3), generator=g) * 0.3
    b = torch.randn(16, generator=g) * 0.1
    ho, wo = H // 2, W // 2
    ref = F.relu(F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2))[:, :, :ho, :wo]
    wp = ops.pack_conv_weight(w.cuda(), dtype, transposed=False)[0]
    y = ops.stem_conv(x.cuda(), wp, b.cuda(), 0, 0, ho, wo, dtype, act=2)
    got = y.view(2, ho, wo, 16).permute(0, 3, 1, 2).float().cpu()
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert float((got - ref).abs().max() / ref.abs().max()) < tol
    t = ops.nchw_to_nhwc(x.cuda(), dtype)
    old = ops.gemm_nt(ops.im2col_ex(t, 2, H, W, 3, 3, 2, 0, 0, ho, wo), wp, bias=b.cuda(), act=2)
    assert float((y.float() - old.float()).abs().max() / old.float().abs().max()) < (1e-5 if dtype == torch.float32 else 2.0 ** -7)
    staged = t.view(2, H, W, 3).permute(0, 3, 1, 2)                    # channels-last compute-dtype image (datastage.image_to_tensor)
    assert torch.equal(ops.stem_conv(staged, wp, b.cuda(), 0, 0, ho, wo, dtype, act=2), y)
