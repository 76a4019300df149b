"""TopFormerModel in train mode (topformer_train.py) against one training step of the REFERENCE (tests/golden/make_topformer_golden.py
... train: train-mode forward, reference OhemCrossEntropyLoss(0, 0.7, 0.1), backward; CPU fp32), and the backward-side kernels of
toptrain.hip against torch autograd on ragged shapes."""
import os
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from semseg_amd.detinit import det_fill_, det_input, det_labels
from _parity import rel, rel_l2, measured  # noqa: E402,F401

VARIANTS = {'T_simple': ('T', 'sum', {'alias': 'simple'}), 'B_refiner': ('B', 'sum', {'alias': 'refiner'}), 'S_simple': ('S', 'sum', {'alias': 'simple'}),
            'T_simple_concat': ('T', 'concat', {'alias': 'simple', 'fusion': 'concat'})}


def make(tag):
    from semseg_amd.topformer import TopFormerModel
    alias, fusion, head = VARIANTS[tag]
    return det_fill_(TopFormerModel(6, alias, 3, 'dot_sum', fusion, frozenset(head.items())))


def _step(net, dtype=None):
    from semseg_amd.loss import OhemCrossEntropyLoss
    from semseg_amd.runtime import set_compute_dtype
    x = det_input('top.xt', (2, 3, 128, 192)).cuda()
    lab = det_labels('top.lab', (2, 128, 192), 6, cell=8).cuda()
    set_compute_dtype(dtype)
    try:
        logits = net(x)
        loss = OhemCrossEntropyLoss(0, 0.7, 0.1)(logits, lab)
        loss.backward()
    finally:
        set_compute_dtype(None)
    return logits, loss


YARD = 4.0          # how many times the reference's own fp32-vs-fp64 deviation the MI355X fp32 path may sit from the fp64 golden
# Per-parameter gradient NORMS get 10 x: a run's deviation from the fp64 golden is ONE draw of the step's noise amplification (which ReLU
# masks flip), shared by every parameter, and the reference's fp32 run is one other draw.  Measured (tools/topformer_train_gradnorm_spread.py):
# replacing the generic fp32 GEMM of the 8 / 16-channel layers by the thread-per-row kernel -- both within 1.6e-7 of float64 on their own
# (tools/narrow_gemm_precision.py) -- moved the median deviation over all parameters from 0.5 x to 3 x the reference's (B_refiner, worst
# parameter 6 x), from 1.0 x to 2.7 x (S_simple) and from 1.3 x to 0.9 x (T_simple).  A wrong chain is off by O(1), not by 1 %.
YARD_GRADNORM = 10.0


@pytest.mark.gpu
@pytest.mark.parametrize('tag', sorted(VARIANTS))
def test_training_step_matches_reference_golden(golden_dir, tag):
    """fp32 step against the reference run in FLOAT64.  The step is poorly conditioned at the golden size (12 token rows feed the
    transformer's train-mode BatchNorms; ReLU / ReLU6 masks flip under 1e-4 noise, which moves gradients by ~sqrt(flipped fraction)):
    the reference's own fp32 run sits 0.7 - 1.2e-4 (logits) and 0.6 - 3e-3 (gradients, rel-L2) from its fp64 self.  The golden file
    records that deviation per quantity (`ref32err.*`, make_topformer_golden.py); the bar here is max(1e-4, YARD x that deviation):
    logits, loss, the L2 norm of EVERY parameter's gradient, sampled full gradients from every part of the network, and the BatchNorm
    running statistics after the step (well conditioned: 1e-4 flat)."""
    g = np.load(os.path.join(golden_dir, f'topformer_{tag}_train.npz'))
    bar = lambda key: max(1e-4, YARD * float(g[key]))
    net = make(tag).cuda().train()
    logits, loss = _step(net)
    assert logits.shape == (2, 6, 128, 192)
    assert rel(logits, g['logits']) < bar('ref32err.logits')
    assert measured(abs(float(loss.detach()) - float(g['loss'])) / abs(float(g['loss'])), 'loss rel err') < bar('ref32err.loss')
    params = dict(net.named_parameters())
    names, norms, nerr = [str(n) for n in g['gradnorm.names']], g['gradnorm.values'], g['ref32err.gradnorm']
    assert set(names) == set(params)
    scale = float(np.max(norms))
    worst = 0.0
    for n, want, ref_dev in zip(names, norms, nerr):
        assert params[n].grad is not None, n
        got = float(params[n].grad.double().norm())
        # per parameter: YARD_GRADNORM x the reference's own deviation for it, or its deviation over all sampled gradients (a single parameter's
        # figure is luck: which ReLU masks flipped); gradients that are zero in exact arithmetic -- the bias in front of another
        # BatchNorm -- hold fp32 noise: the absolute term
        allowed = YARD_GRADNORM * max(float(ref_dev), max(1e-4, float(g['ref32err.grad_l2_all'])) * want) + 1e-6 * scale
        assert abs(got - want) < allowed, (n, got, want, float(ref_dev))
        worst = max(worst, abs(got - want) / allowed)
    measured(worst, 'gradient-norm deviation / allowed (all parameters)') < 1.0
    full = [k for k in g.files if k.startswith('grad.')]
    assert len(full) > 20
    got_all, want_all = [], []
    for k in full:
        want = torch.from_numpy(g[k])
        got = params[k[5:]].grad.cpu()
        got_all.append(got.flatten().double()), want_all.append(want.flatten().double())
        if float(want.abs().max()) < 1e-9 * scale:                      # zero in exact arithmetic
            assert float(got.abs().max()) < 1e-5 * scale, k
        else:
            assert rel(got, want) < bar('ref32err.' + k), k
    got_all, want_all = torch.cat(got_all), torch.cat(want_all)
    assert measured(float((got_all - want_all).norm() / want_all.norm()), 'rel-L2, sampled gradients as one vector') < bar('ref32err.grad_l2_all')
    sd = net.state_dict()
    for k in (k for k in g.files if k.startswith('after.')):
        if k.endswith('num_batches_tracked'):
            assert int(sd[k[6:]]) == int(g[k]), k
        else:
            assert rel(sd[k[6:]], g[k]) < 1e-4, k


def _run_piece(T, build, inputs, seed=3):
    """One tape piece in compute dtype T: `build(tape, *vars)` -> output _V; a fixed random gradient is pushed back.
    -> (output, input gradients, {parameter: gradient})"""
    from semseg_amd import topformer_train as tt
    tape = tt._Tape(T, True)
    vs = [tt._V(t.to(T).contiguous()) for t in inputs]
    out = build(tape, *vs)
    gen = torch.Generator().manual_seed(seed)
    out.g = torch.randn(out.t.shape, generator=gen).cuda().to(T)
    tape.run_backward()
    return out.t.float(), [v.g.float() for v in vs], {p: g.float() for p, g in tape.pgrads.items()}


@pytest.mark.gpu
@pytest.mark.parametrize('piece', ['ir_block_s2', 'ir_block_res', 'attention', 'ffn', 'injection', 'simple_head', 'refiner_head'])
def test_training_pieces_bf16_vs_fp32(piece):
    """bf16 against fp32 per building block, forward and backward, on the same (bf16-representable) inputs and output gradient.
    Whole-model comparisons are meaningless in train mode at random init: the network amplifies any perturbation ~2000x from the image
    to the logits (each train-mode BatchNorm re-normalises, each extractor block doubles it: tools/topformer_train_layer_drift.py), in
    the reference as here, so a bf16 rounding of the stem output alone moves the logits by O(1).  Bars: 2e-2 rel-L2 on outputs.
    Gradients of a ReLU / ReLU6 block in bf16 sit 3 - 6e-2 from fp32 whatever the kernel: a unit whose normalised pre-activation is
    within one bf16 rounding of zero (|y| < ~4e-3, ~0.2 % of the units) flips its mask, and flipping a fraction p of the units moves the
    gradient by ~sqrt(p); the un-scaled softmax of the attention block (logits of size ~10 from bf16 q, k) adds its own 4e-2.  So:
    1e-1 on input and parameter gradients (1.5e-1 attention), which still catches any wrong term."""
    from semseg_amd import topformer_train as tt
    from semseg_amd.topformer import TopFormerModel
    head = 'refiner' if piece == 'refiner_head' else 'simple'
    model = det_fill_(TopFormerModel(6, 'B', head_cfg=frozenset({'alias': head}.items()))).cuda().train()
    net = model.backbone.net
    B = 2
    gen = torch.Generator().manual_seed(11)
    rnd = lambda *s: torch.randn(*s, generator=gen).cuda().bfloat16().float()
    if piece in ('ir_block_s2', 'ir_block_res'):
        blk = net.tpm.net.staging_module.stage_2.resblock_1 if piece == 'ir_block_s2' else net.tpm.net.staging_module.stage_2.resblock_2
        H, W = 24, 40
        inputs = [rnd(B * H * W, blk.net.conv_in.inc)]

        def build(tape, y):
            z = tt.conv_bn(tape, y, blk.net.conv_in)
            z, _, _ = tt.dw_bn(tape, z, blk.net.conv_dw, B, H, W)
            return tt.conv_bn(tape, z, blk.net.conv_out, res=y if blk.residual_connect else None)
        assert blk.residual_connect == (piece == 'ir_block_res')
    elif piece == 'attention':
        n = 8 * 12
        inputs = [rnd(B * n, net.embed_dim)]
        build = lambda tape, tok: tt.attention(tape, tok, net.trans.transformer_sequence.extractor_1.attn, B, n, net.num_heads, net.key_dim, net.d)
    elif piece == 'ffn':
        hg, wg = 8, 12
        inputs = [rnd(B * hg * wg, net.embed_dim)]
        m = net.trans.transformer_sequence.extractor_2.mlp.net

        def build(tape, tok):
            z = tt.conv_bn(tape, tok, m.preconv)
            z, _, _ = tt.dw_bn(tape, z, m.dwconv, B, hg, wg)
            return tt.conv_bn(tape, z, m.postconv, res=tok)
    elif piece == 'injection':
        (hg, wg), (ph, pw) = (4, 6), (32, 48)
        c = net.channel_splits[1]
        inputs = [rnd(B * ph * pw, c), rnd(B * hg * wg, c)]
        m = net.post_stages.stage_2

        def build(tape, f, xg):
            local = tt.conv_bn(tape, f, m.local_embedding)
            ga, ge = tt.conv_bn(tape, xg, m.global_act), tt.conv_bn(tape, xg, m.global_embedding)
            out = tt._V(tt.ops.inject_fuse(local.t, ga.t, ge.t, B, ph, pw, hg, wg))

            def bwd():
                g = out.g
                tape.acc(local, tt.ops.mul(g, tt.ops.bilinear_fwd(ga.t, B, hg, wg, ph, pw)))
                tape.acc(ga, tt.ops.bilinear_bwd(tt.ops.mul(g, local.t), B, hg, wg, ph, pw))
                tape.acc(ge, tt.ops.bilinear_bwd(g, B, hg, wg, ph, pw))
            tape.push(bwd)
            return out
    else:
        sizes = [(32, 48), (16, 24), (8, 12), (4, 6)]
        # (features with per-image structure: SimAM and the GroupNorms divide by the per-image spread)
        inputs = [rnd(B * h * w, 256) + 0.5 * rnd(1, 256) for h, w in sizes]
        fn = tt.refiner_head if piece == 'refiner_head' else tt.simple_head
        build = lambda tape, *fs: fn(tape, model, [(f, h, w) for f, (h, w) in zip(fs, sizes)], B, 32, 48)
    o32, gi32, pg32 = _run_piece(torch.float32, build, inputs)
    o16, gi16, pg16 = _run_piece(torch.bfloat16, build, inputs)
    loose = piece == 'refiner_head'          # (two per-image normalisations behind a sigmoid: ~3x the error of their input, see test_topformer_gpu.py)
    assert rel_l2(o16, o32) < (6e-2 if loose else 2e-2)
    gbar = 1.5e-1 if piece == 'attention' else 1e-1
    for a, b in zip(gi16, gi32):
        assert rel_l2(a, b) < gbar
    assert set(pg16) == set(pg32) and len(pg32) >= 3
    cat = lambda d: torch.cat([d[p].flatten() for p in pg32])
    assert rel_l2(cat(pg16), cat(pg32)) < gbar


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['B_refiner', 'T_simple'])
def test_training_step_bf16_runs(tag):
    """The whole bf16 step: dtype, finiteness and a loss in the range of the fp32 one (see test_training_pieces_bf16_vs_fp32 for why
    nothing tighter can be asked of a train-mode forward at random init)."""
    net = make(tag).cuda().train()
    logits, loss = _step(net, torch.bfloat16)
    assert logits.dtype == torch.bfloat16 and bool(torch.isfinite(logits.float()).all())
    assert 1.0 < float(loss.detach()) < 3.5                             # ln 6 = 1.79: an untrained six-class model
    for n, p in net.named_parameters():
        assert p.grad is not None and p.grad.dtype == torch.float32 and bool(torch.isfinite(p.grad).all()), n


@pytest.mark.gpu
def test_train_mode_under_no_grad_and_second_step():
    """A train-mode forward under no_grad keeps nothing for a backward but still updates the running statistics (as nn.BatchNorm2d
    does); a second step on the same module works (the tape is per call) and eval mode afterwards uses the updated statistics."""
    net = make('T_simple').cuda().train()
    x = det_input('top.xt', (2, 3, 128, 192)).cuda()
    bn = net.backbone.net.tpm.net.stem.net.norm
    with torch.no_grad():
        a = net(x)
    assert not a.requires_grad and int(bn.num_batches_tracked) == 1
    b = net(x)
    assert b.requires_grad and int(bn.num_batches_tracked) == 2
    assert torch.equal(a, b)                                        # batch statistics: the running buffers do not enter a train-mode forward
    b.float().square().mean().backward()
    g1 = net.classifier.net.p_conv.conv.weight.grad.clone()
    net.zero_grad()
    c = net(x)
    c.float().square().mean().backward()
    assert torch.equal(net.classifier.net.p_conv.conv.weight.grad, g1)      # fixed-order reductions: a step is bitwise reproducible
    e = net.eval()(x)
    assert e.shape == a.shape and bool(torch.isfinite(e).all()) and not torch.equal(e, a)


# ---------------------------------------------------------------------------------- kernels of toptrain.hip against torch autograd
@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('C,k,s,hw', [(16, 3, 1, (12, 20)), (64, 3, 2, (16, 24)), (48, 5, 2, (14, 10)), (96, 5, 1, (9, 7)), (416, 3, 1, (2, 3)), (24, 5, 2, (8, 8))])
def test_depthwise_strided_gradients_vs_torch(dtype, C, k, s, hw):
    """ss_dwconvk_dgrad / ss_dwconvk_wgrad_s (stride 1 / 2, AutoPadding: smaller half first) against autograd of F.conv2d on the same
    (rounded) operands; fp32 1e-5, bf16 one output rounding."""
    from semseg_amd import ops
    from semseg_amd.topformer import auto_pad
    B, (H, W) = 2, hw
    gen = torch.Generator().manual_seed(C + k + s)
    x = torch.randn(B, C, H, W, generator=gen).to(dtype).float().requires_grad_(True)
    w = (torch.randn(C, 1, k, k, generator=gen) * 0.3).requires_grad_(True)
    b = torch.zeros(C, requires_grad=True)
    p0, p1 = auto_pad(k, s)
    Ho, Wo = H // s, W // s
    y = F.conv2d(F.pad(x, (p0, p1, p0, p1)), w, b, stride=s, groups=C)[:, :, :Ho, :Wo]
    gy = torch.randn(B, C, Ho, Wo, generator=gen).to(dtype).float()
    y.backward(gy)
    tok = lambda t: t.detach().permute(0, 2, 3, 1).reshape(-1, C).contiguous().cuda().to(dtype)
    wkkc = ops.transpose_cast(w.detach().reshape(C, k * k).contiguous().cuda(), torch.float32)
    dx = ops.dwconvk_dgrad(tok(gy), wkkc, B, H, W, k, s, p0, p0, Ho, Wo)
    tol = 1e-5 if dtype == torch.float32 else 6e-3
    assert rel(dx.view(B, H, W, C).permute(0, 3, 1, 2), x.grad) < tol
    dwdb = ops.dwconvk_wgrad_s(tok(gy), tok(x), B, H, W, k, s, p0, p0, Ho, Wo)
    assert rel(dwdb[:k * k].t().reshape(C, 1, k, k), w.grad) < 2e-5
    assert rel(dwdb[k * k], b.grad) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_avgpool_act_mul_gradients_vs_torch(dtype):
    from semseg_amd import ops
    B, C, H, W, win = 2, 48, 16, 24, 8
    gen = torch.Generator().manual_seed(5)
    g = torch.randn(B * (H // win) * (W // win), C + 16, generator=gen).cuda().to(dtype)
    dx = ops.avgpool_bwd(g[:, 8:8 + C], B, H, W, win)
    gi = g[:, 8:8 + C].float().view(B, H // win, W // win, C).permute(0, 3, 1, 2)
    want = F.interpolate(gi, scale_factor=win, mode='nearest') / (win * win)
    assert rel(dx.view(B, H, W, C).permute(0, 3, 1, 2), want) < (1e-6 if dtype == torch.float32 else 4e-3)
    y = (torch.randn(1000, 64, generator=gen) * 4).cuda().to(dtype).float()       # (rounded first: 5.99 must not become 6.0 between the two sides)
    dy = torch.randn(1000, 64, generator=gen).cuda().to(dtype)
    for code, fn in ((2, F.relu), (3, F.relu6), (5, torch.sigmoid)):
        xin = y.clone().requires_grad_(True)
        out = fn(xin)
        out.backward(dy.float())
        got = ops.act_bwd(dy, out.detach().to(dtype), code)
        if dtype == torch.float32:
            assert rel(got, xin.grad) < 1e-6, code
        else:
            assert rel_l2(got, xin.grad) < 1e-2, code
    a, b = torch.randn(999, 8, generator=gen).cuda().to(dtype), torch.randn(999, 8, generator=gen).cuda().to(dtype)
    assert rel(ops.mul(a, b), a.float() * b.float()) < (1e-7 if dtype == torch.float32 else 4e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_simam_backward_vs_autograd(dtype):
    """ss_simam_bwd (gradient through the per-image mean and variance and to the learnable k) against autograd of modules/heads.py:27-39
    restated in torch, 3 images of 35 x 9 pixels."""
    from semseg_amd import ops
    b, n, C = 3, 35 * 9, 64
    gen = torch.Generator().manual_seed(1)
    x = (torch.randn(b * n, C, generator=gen) * 2 + torch.randn(C, generator=gen)).cuda().to(dtype)
    dy = torch.randn(b * n, C, generator=gen).cuda().to(dtype)
    k = torch.tensor([0.01], device='cuda', requires_grad=True)
    xi = x.float().view(b, n, C).requires_grad_(True)
    d2 = (xi - xi.mean(1, keepdim=True)).pow(2)
    y = xi * (d2 / (4 * (k + d2.sum(1, keepdim=True) / (n - 1))) + 0.5).sigmoid()
    y.backward(dy.float().view(b, n, C))
    mom = ops.image_moments(x, b)
    dk = torch.zeros(1, device='cuda')
    dx = ops.simam_bwd(dy, x, mom, k.detach(), b, dk)
    assert rel(dx.view(b, n, C), xi.grad) < (2e-5 if dtype == torch.float32 else 8e-3)
    assert rel(dk, k.grad) < 1e-4
    dk2 = dk.clone()
    ops.simam_bwd(dy, x, mom, k.detach(), b, dk2)
    assert rel(dk2, 2 * k.grad) < 1e-4                                   # accumulate_dk


@pytest.mark.gpu
@pytest.mark.parametrize('groups,cg,pitch,pre', [(4, 19, 24, 5), (4, 6, 8, 5), (6, 1, 1, 0), (10, 1, 1, 0), (2, 5, 8, 0)])
def test_groupnorm_fwd_bwd_vs_torch(groups, cg, pitch, pre):
    """ss_groupnorm_fwd / _bwd on the padded channels-last layout (optional sigmoid in front) against nn.GroupNorm + autograd."""
    from semseg_amd import ops
    b, n = 3, 35 * 9
    ncol = (max(groups * pitch, (groups - 1) * pitch + cg) + 7) // 8 * 8
    gen = torch.Generator().manual_seed(groups * 100 + cg)
    val = torch.randn(b * n, groups, cg, generator=gen).cuda().requires_grad_(True)
    u = torch.full((b * n, ncol), 0.37, device='cuda')                  # padding columns hold junk: they must be ignored and written as zero
    cols = (torch.arange(groups)[:, None] * pitch + torch.arange(cg)[None]).flatten().cuda()
    u[:, cols] = val.detach().reshape(b * n, -1)
    gn = torch.nn.GroupNorm(groups, groups * cg).cuda()
    with torch.no_grad():
        gn.weight.normal_(1, 0.2), gn.bias.normal_(0, 0.2)
    f = val.view(b, n, groups * cg)
    f = f.sigmoid() if pre else f
    want = gn(f.permute(0, 2, 1).reshape(b, groups * cg, n, 1)).view(b, groups * cg, n).permute(0, 2, 1)
    dy = torch.randn(b * n, ncol, generator=gen).cuda()
    want.backward(dy[:, cols].view(b, n, -1))
    y, stats = ops.groupnorm_fwd(u, b, groups, cg, pitch, gn.eps, gn.weight.detach(), gn.bias.detach(), pre_act=pre)
    assert rel(y[:, cols].view(b, n, -1), want) < 1e-5
    pad = torch.ones(ncol, dtype=torch.bool, device='cuda')
    pad[cols] = False
    assert float(y[:, pad].abs().max()) == 0.0 if bool(pad.any()) else True
    dx, dgb = ops.groupnorm_bwd(dy, u, stats, gn.weight.detach(), b, groups, cg, pitch, pre_act=pre)
    assert rel(dx[:, cols].view(b * n, groups, cg), val.grad) < 2e-5
    assert rel(dgb[0], gn.weight.grad) < 2e-5 and rel(dgb[1], gn.bias.grad) < 2e-5
    if bool(pad.any()):
        assert float(dx[:, pad].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('C,r,hg,wg', [(256, 16, 2, 3), (128, 8, 4, 6), (192, 4, 3, 5), (64, 2, 8, 4), (8, 2, 1, 1), (256, 6, 2, 2)])
def test_injection_backward_in_one_pass(dtype, C, r, hg, wg):
    """ss_inject_bwd (dlocal, dgact, dgemb from one read of g and local) against autograd of local * up(gact) + up(gemb) in torch fp32
    (F.interpolate bilinear, align_corners=False) and against the five launches it replaces; dlocal is bit-identical to those."""
    from semseg_amd import ops
    B, hl, wl = 2, r * hg, r * wg
    gen = torch.Generator().manual_seed(C + r)
    mk = lambda n: torch.randn(n, C, generator=gen).cuda().to(dtype)
    g, local, gact = mk(B * hl * wl), mk(B * hl * wl), mk(B * hg * wg)
    dlocal, dgact, dgemb = ops.inject_bwd(g, local, gact, B, hl, wl, hg, wg)
    nchw = lambda t, h, w: t.float().view(B, h, w, C).permute(0, 3, 1, 2)
    lo, ga, ge = nchw(local, hl, wl).requires_grad_(True), nchw(gact, hg, wg).requires_grad_(True), torch.zeros(B, C, hg, wg, device='cuda', requires_grad=True)
    up = lambda t: F.interpolate(t, size=(hl, wl), mode='bilinear', align_corners=False)
    (lo * up(ga) + up(ge)).backward(nchw(g, hl, wl))
    tol = 2e-5 if dtype == torch.float32 else 8e-3
    assert rel(nchw(dlocal, hl, wl), lo.grad) < tol
    assert rel(nchw(dgact, hg, wg), ga.grad) < tol
    assert rel(nchw(dgemb, hg, wg), ge.grad) < tol
    old = ops.mul(g, ops.bilinear_fwd(gact, B, hg, wg, hl, wl))
    assert torch.equal(dlocal, old)
    assert rel(dgemb, ops.bilinear_bwd(g, B, hg, wg, hl, wl)) < tol
