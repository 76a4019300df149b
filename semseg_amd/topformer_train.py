"""TopFormerModel in train mode on the MI355X kernels: what train.py:45-52 + :306-330 of the reference runs per step (train-mode
forward with batch-statistic BatchNorm and running-statistic updates, then the backward down to every parameter).

The step is ONE autograd node (``_TopFormerTrainFn``) around a tape of kernel launches over channels-last token matrices: the forward
appends, per layer, a closure that knows how to push the layer's output gradient to its inputs and parameters; the backward runs the
closures in reverse.  PyTorch allocates and hands the gradients to the optimizer side (``runtime.deliver_grads``: straight into
engine.FlatParams' flat buffer when the model is registered there); every FLOP is a kernel of libsemseg_hip.so.

Layer by layer (reference file:line):
* conv 1x1 + BatchNorm (+ ReLU / ReLU6) (modules/blocks.py:17-198 'CNA'): ss_gemm_nt -> ss_bn_stats (updates running_mean / running_var
  / num_batches_tracked in place) -> ss_bn_apply (activation and residual in the same pass); backward ss_bn_bwd (the activation mask from
  the recomputed pre-activation) -> ss_gemm_tn / ss_gemm_nt.
* depthwise k x k, stride 1 / 2, AutoPadding (modules/topformer.py:166-184, modules/layers.py:146-164): ss_dwconvk_fwd; backward
  ss_dwconvk_dgrad + ss_dwconvk_wgrad_s.
* stem 3x3 / stride 2 (modules/topformer.py:1023-1033): ss_im2col_ex + GEMM (the weight gradient is one ss_gemm_tn on the patch matrix).
* PyramidPoolAgg (:527-552): ss_avgpool into channel slices of the token matrix; backward ss_avgpool_bwd.
* AttentionBlock (:330-427; q, k, v each conv + BN, un-scaled softmax(q^T k), projection ReLU6 -> conv -> BN 'ACN'): the MiT flash kernels
  with q / k zero-padded to the value head width (padding channels: zero weight rows and gamma = beta = 0, so they stay exactly zero
  through BatchNorm and receive exactly zero gradient).
* FFN (:236-327): conv-BN -> depthwise 3x3 (bias) -BN-ReLU6 -> conv-BN, residual in the last BatchNorm's apply pass.
* InjectionDotSum (:640-673): ss_inject_fuse forward; backward ss_inject_bwd (one pass over the fine map: the three gradients from one read
  of g and local; other geometries: ss_bilinear_fwd + ss_mul + ss_bilinear_bwd).
* SimpleHead (modules/heads.py:120-170) and the classifier (core/models.py:103-115): UniRes + Sum / Concat, conv-BN-ReLU6, then the 1x1
  classifier BEFORE the final bilinear resize (both linear; exact in real arithmetic).
* RefinerHead (modules/heads.py:173-269): SimAM (ss_image_moments + ss_simam_apply / ss_simam_bwd), the grouped 1x1 conv per stage
  before the resize, sigmoid + GroupNorm(n_heads) (ss_groupnorm_fwd / _bwd with pre_act), the view + Conv3d(n_heads -> 1) as one small
  fixed-pattern matrix, GroupNorm(num_classes groups), classifier; fp32 from the group conv on.
DropPath rate is 0 in every reference config (modules/topformer.py:1121-1378) and there is no Dropout on this path.
"""
import os

import torch

from . import ops, weightprep
from .runtime import GradAwareFunction, compute_dtype_for, deliver_grads, expect_grads, recording
from .topformer import RELU, RELU6, RefinerHead, SimpleHead, auto_pad

F32 = torch.float32


class _V:
    """A tensor on the tape with its gradient slot."""
    __slots__ = ('t', 'g', 'pending')

    def __init__(self, t):
        self.t, self.g, self.pending = t, None, None


class _Tape:
    def __init__(self, T, record):
        self.T, self.record, self.steps, self.pgrads, self._zeros = T, record, [], {}, {}

    def push(self, fn):
        if self.record:
            self.steps.append(fn)

    @staticmethod
    def acc(v, g):
        v.g = g if v.g is None else ops.add(v.g, g)

    def pg(self, p, g):
        """gradient of parameter p (fp32, shaped like p; a parameter used once per step, so no accumulation is needed)"""
        g = g.reshape(p.shape)
        old = self.pgrads.get(p)
        self.pgrads[p] = g if old is None else ops.add(old.contiguous(), g.contiguous())

    def zeros(self, n, dev):
        z = self._zeros.get((n, dev))
        if z is None:
            z = self._zeros[(n, dev)] = torch.zeros(n, dtype=F32, device=dev)
        return z

    def run_backward(self):
        for fn in reversed(self.steps):
            fn()
        self.steps = []

    # ------------------------------------------------------------------ operand preparation
    def wmat(self, w):
        """conv weight [O, I, 1, 1] (a Parameter: the compute-dtype copy is remembered across steps and refreshed with all others in one
        launch, weightprep.py) or a derived fp32 [O, I] matrix -> GEMM operand [O, I] in the compute dtype"""
        if isinstance(w, torch.nn.Parameter):
            return weightprep.cast(w, self.T)
        w = w.contiguous()
        return w if self.T == F32 else ops.cast(w, self.T)

    def wmat_t(self, w):
        """-> [I, O] in the compute dtype (the data-gradient GEMM's operand)"""
        if isinstance(w, torch.nn.Parameter):
            return weightprep.transpose(w, w.shape[0], w.numel() // w.shape[0], self.T)
        return ops.transpose_cast(w.contiguous(), self.T)


def _bn_args(bn):
    if bn.momentum is None or not bn.track_running_stats or not bn.affine:
        raise NotImplementedError('TopFormer training on the MI355X path: BatchNorm must be affine with running statistics and a fixed momentum')
    if not bn.training:
        raise NotImplementedError('TopFormer training on the MI355X path: a BatchNorm in eval mode inside a model in train mode (frozen statistics) is not '
                                  'built; call .train() / .eval() on the whole model')
    return bn.weight.detach(), bn.bias.detach(), float(bn.eps), float(bn.momentum)


def _conv_bn_fwd(x, w, bn, res=None, act=0):
    """1x1 conv and the batch statistics of its output from one pass (the GEMM's copy-out, where the tile kernels take the shape)"""
    gamma, beta, eps, mom = _bn_args(bn)
    z, stats = ops.gemm_nt_bn_stats(x, w, eps=eps, momentum=mom, running_mean=bn.running_mean, running_var=bn.running_var, nbt=bn.num_batches_tracked)
    return z, ops.bn_apply(z, stats, gamma, beta, act=act, res=res), stats


def _bn_fwd(z, bn, res=None, out=None, act=0):
    gamma, beta, eps, mom = _bn_args(bn)
    stats = ops.bn_stats(z, eps, mom, bn.running_mean, bn.running_var, bn.num_batches_tracked)
    return ops.bn_apply(z, stats, gamma, beta, act=act, res=res, out=out), stats


def conv_bn(tape, x, unit, res=None, need_dx=True, defer_dx=False):
    """1x1 conv + BatchNorm + unit.act on a token matrix (x.t may be a channel slice); res: added after the norm.
    defer_dx: the input gradient is left to the owner of x (x.pending collects (dz, W^T))."""
    conv, bn = unit.net.p_conv.conv, unit.net.norm
    if conv.bias is not None or unit.k != 1 or unit.order != 'CNA':
        raise NotImplementedError('conv_bn: 1x1 bias-free conv-norm-act units only')
    w2 = conv.weight
    z, y, stats = _conv_bn_fwd(x.t, tape.wmat(w2), bn, res=res.t if res is not None else None, act=unit.act)      # ReLU / ReLU6 inside the normalisation pass
    out = _V(y)

    def bwd():
        g = out.g
        if res is not None:
            tape.acc(res, g)
        dz, dgb = ops.bn_bwd(g, z, stats, bn.weight.detach(), bn.bias.detach(), act=unit.act)      # mask from the recomputed pre-activation
        tape.pg(bn.weight, dgb[0]), tape.pg(bn.bias, dgb[1])
        tape.pg(conv.weight, ops.gemm_tn(dz, x.t))
        if defer_dx:
            x.pending.append((dz, tape.wmat_t(w2)))
        elif need_dx:
            tape.acc(x, ops.gemm_nt(dz, tape.wmat_t(w2)))
    tape.push(bwd)
    return out


def dw_bn(tape, x, unit, B, H, W):
    """depthwise k x k (stride 1 / 2, AutoPadding) + BatchNorm + unit.act -> (out, Ho, Wo)"""
    conv, bn = unit.net.p_conv.conv, unit.net.norm
    C, k, s = conv.weight.shape[0], unit.k, unit.stride
    wkkc = weightprep.transpose(conv.weight, C, k * k, F32)
    bias = conv.bias.detach() if conv.bias is not None else tape.zeros(C, x.t.device)
    p0, _ = auto_pad(k, s)
    Ho, Wo = H // s, W // s
    z = ops.dwconvk_fwd(x.t, wkkc, bias, B, H, W, k, s, p0, p0, Ho, Wo, act=0)
    y, stats = _bn_fwd(z, bn, act=unit.act)
    out = _V(y)

    def bwd():
        dz, dgb = ops.bn_bwd(out.g, z, stats, bn.weight.detach(), bn.bias.detach(), act=unit.act)
        tape.pg(bn.weight, dgb[0]), tape.pg(bn.bias, dgb[1])
        if s == 1 and p0 == k // 2 and os.environ.get('SS_TOP_WGRAD', 'tile') == 'tile':   # stride 1, symmetric padding: the LDS-tiled kernel of the ConvNeXt path
            dwdb = ops.dwconvk_wgrad(dz, x.t, B, H, W, k)
        else:
            dwdb = ops.dwconvk_wgrad_s(dz, x.t, B, H, W, k, s, p0, p0, Ho, Wo)
        tape.pg(conv.weight, ops.transpose_cast(dwdb[:k * k], F32))
        if conv.bias is not None:
            tape.pg(conv.bias, dwdb[k * k])
        if s == 1 and p0 == k // 2:                     # stride 1: the data gradient is the forward kernel (LDS-tiled) with the taps reversed
            wflip = weightprep.transpose(conv.weight, C, k * k, F32, reverse_cols=True)
            tape.acc(x, ops.dwconvk_fwd(dz, wflip, tape.zeros(C, dz.device), B, H, W, k, 1, k - 1 - p0, k - 1 - p0, H, W, act=0))
        else:
            tape.acc(x, ops.dwconvk_dgrad(dz, wkkc, B, H, W, k, s, p0, p0, Ho, Wo))
    tape.push(bwd)
    return out, Ho, Wo


def stem(tape, x, unit):
    """3x3 / stride 2 conv + BN + ReLU straight from the fp32 NCHW image -> (tokens, h, w); no gradient to the image."""
    conv, bn = unit.net.p_conv.conv, unit.net.norm
    B, cin, H, W = x.shape
    pt, _ = auto_pad(3, 2)
    h, w = H // 2, W // 2
    t = ops.nchw_to_nhwc(x.contiguous().float(), tape.T)
    col = ops.im2col_ex(t, B, H, W, cin, 3, 2, pt, pt, h, w)
    wp = weightprep.pack(conv.weight, tape.T, transposed=False)
    z, y, stats = _conv_bn_fwd(col, wp, bn, act=unit.act)
    out = _V(y)

    def bwd():
        dz, dgb = ops.bn_bwd(out.g, z, stats, bn.weight.detach(), bn.bias.detach(), act=unit.act)
        tape.pg(bn.weight, dgb[0]), tape.pg(bn.bias, dgb[1])
        tape.pg(conv.weight, ops.unpack_conv_wgrad(ops.gemm_tn(dz, col).contiguous(), conv.weight.shape[0], cin, 3))
    tape.push(bwd)
    return out, h, w


def _pad_heads(t, nh, kd, d, fill=0.0):
    out = torch.full((nh, d) + tuple(t.shape[1:]), fill, dtype=t.dtype, device=t.device)
    out[:, :kd] = t.reshape((nh, kd) + tuple(t.shape[1:]))
    return out.reshape((nh * d,) + tuple(t.shape[1:]))


def _unpad_heads(t, nh, kd, d):
    return t.reshape((nh, d) + tuple(t.shape[1:]))[:, :kd].reshape((nh * kd,) + tuple(t.shape[1:])).contiguous()


class _PaddedBN:
    """BatchNorm of a q / k branch over channels padded per head from key_dim to the value head width.  The padding channels carry
    gamma = beta = 0 and see z = 0: they stay 0 and take no gradient; the real channels' running statistics are copied back."""

    def __init__(self, bn, nh, kd, d):
        self.bn, self.geo = bn, (nh, kd, d)
        gamma, beta, self.eps, self.mom = _bn_args(bn)
        self.gamma, self.beta = _pad_heads(gamma, nh, kd, d), _pad_heads(beta, nh, kd, d)

    def fwd(self, z, out=None):
        bn, (nh, kd, d) = self.bn, self.geo
        rm, rv = _pad_heads(bn.running_mean, nh, kd, d), _pad_heads(bn.running_var, nh, kd, d, 1.0)
        self.stats = ops.bn_stats(z, self.eps, self.mom, rm, rv, bn.num_batches_tracked)
        bn.running_mean.copy_(_unpad_heads(rm, nh, kd, d)), bn.running_var.copy_(_unpad_heads(rv, nh, kd, d))
        return ops.bn_apply(z, self.stats, self.gamma, self.beta, out=out)

    def bwd(self, tape, dy, z, out=None):
        dz, dgb = ops.bn_bwd(dy, z, self.stats, self.gamma, self.beta, out=out)
        nh, kd, d = self.geo
        tape.pg(self.bn.weight, _unpad_heads(dgb[0], nh, kd, d)), tape.pg(self.bn.bias, _unpad_heads(dgb[1], nh, kd, d))
        return dz


def attention(tape, tok, attn, B, n, nh, kd, d):
    """tok + projection(softmax(q^T k) v) (modules/topformer.py:330-427, SemanticsExtractorBlock :470-487 with drop_path 0)"""
    mb, pj = attn.multi_branch, attn.projection
    T, Cq = tape.T, nh * d
    cw = lambda u: u.net.p_conv.conv.weight.detach().reshape(u.net.p_conv.conv.weight.shape[0], -1)
    wq, wk, wv = _pad_heads(cw(mb.branch_q), nh, kd, d), _pad_heads(cw(mb.branch_k), nh, kd, d), cw(mb.branch_v)
    x = tok.t
    rows = x.shape[0]
    zq = ops.gemm_nt(x, tape.wmat(wq))
    zkv = torch.empty((rows, 2 * Cq), dtype=T, device=x.device)
    ops.gemm_nt(x, tape.wmat(wk), out=zkv[:, :Cq]), ops.gemm_nt(x, tape.wmat(wv), out=zkv[:, Cq:])
    bq, bk = _PaddedBN(mb.branch_q.net.norm, nh, kd, d), _PaddedBN(mb.branch_k.net.norm, nh, kd, d)
    bnv = mb.branch_v.net.norm
    q = bq.fwd(zq)
    kv = torch.empty_like(zkv)
    bk.fwd(zkv[:, :Cq], out=kv[:, :Cq])
    _, stats_v = _bn_fwd(zkv[:, Cq:], bnv, out=kv[:, Cq:])
    ao, lse = ops.mit_attn_fwd(q, kv, B, n, n, nh, scale=1.0)
    a = ops.act(ao, pj.act)
    wp = pj.net.p_conv.conv.weight
    bnp = pj.net.norm
    zp, y, stats_p = _conv_bn_fwd(a, tape.wmat(wp), bnp, res=x)
    out = _V(y)

    def bwd():
        g = out.g
        dzp, dgb = ops.bn_bwd(g, zp, stats_p, bnp.weight.detach(), bnp.bias.detach())
        tape.pg(bnp.weight, dgb[0]), tape.pg(bnp.bias, dgb[1])
        tape.pg(pj.net.p_conv.conv.weight, ops.gemm_tn(dzp, a))
        dao = ops.act_bwd(ops.gemm_nt(dzp, tape.wmat_t(wp)), a, pj.act)
        dq, dkv = ops.mit_attn_bwd(q, kv, ao, dao, lse, B, n, n, nh, scale=1.0)
        dzq = bq.bwd(tape, dq, zq)
        dzkv = torch.empty_like(zkv)
        bk.bwd(tape, dkv[:, :Cq], zkv[:, :Cq], out=dzkv[:, :Cq])
        _, dgbv = ops.bn_bwd(dkv[:, Cq:], zkv[:, Cq:], stats_v, bnv.weight.detach(), bnv.bias.detach(), out=dzkv[:, Cq:])
        tape.pg(bnv.weight, dgbv[0]), tape.pg(bnv.bias, dgbv[1])
        tape.pg(mb.branch_q.net.p_conv.conv.weight, _unpad_heads(ops.gemm_tn(dzq, x), nh, kd, d))
        tape.pg(mb.branch_k.net.p_conv.conv.weight, _unpad_heads(ops.gemm_tn(dzkv[:, :Cq], x), nh, kd, d))
        tape.pg(mb.branch_v.net.p_conv.conv.weight, ops.gemm_tn(dzkv[:, Cq:], x))
        dx = ops.gemm_nt(dzq, tape.wmat_t(wq), res=g)                       # the residual path's gradient rides in the epilogue
        tape.acc(tok, ops.gemm_nt(dzkv, tape.wmat_t(torch.cat([wk, wv])), res=dx))
    tape.push(bwd)
    return out


def backbone(tape, net, x):
    """TopFormerModule.forward (modules/topformer.py:1036-1117) in train mode -> [(injected tokens _V, h, w)] per stage"""
    B = x.shape[0]
    y, h, w = stem(tape, x, net.tpm.net.stem)
    pyr = []
    for stage in net.tpm.net.staging_module:
        for blk in stage:
            z = conv_bn(tape, y, blk.net.conv_in) if hasattr(blk.net, 'conv_in') else y
            z, ho, wo = dw_bn(tape, z, blk.net.conv_dw, B, h, w)
            y = conv_bn(tape, z, blk.net.conv_out, res=y if blk.residual_connect else None)
            h, w = ho, wo
        pyr.append((y, h, w))
    hg, wg = h // net.c2t_stride, w // net.c2t_stride
    if any(ph % hg or pw % wg or ph // hg != pw // wg for _, ph, pw in pyr):
        raise ValueError(f'TopFormer: input {x.shape[2]}x{x.shape[3]} must be a multiple of 64 (adaptive pooling windows must be exact)')
    tok0 = _V(torch.empty((B * hg * wg, net.embed_dim), dtype=tape.T, device=x.device))
    offs, off = [], 0
    for (f, ph, pw), c in zip(pyr, net.channel_splits):
        ops.avgpool(f.t, B, ph, pw, ph // hg, out=tok0.t[:, off:off + c])
        offs.append((off, c))
        off += c

    def pool_bwd():
        for (f, ph, pw), (o, c) in zip(pyr, offs):
            tape.acc(f, ops.avgpool_bwd(tok0.g[:, o:o + c], B, ph, pw, ph // hg))
    tape.push(pool_bwd)
    tok = tok0
    for ex in net.trans.transformer_sequence:
        tok = attention(tape, tok, ex.attn, B, hg * wg, net.num_heads, net.key_dim, net.d)
        m = ex.mlp.net
        z = conv_bn(tape, tok, m.preconv)
        z, _, _ = dw_bn(tape, z, m.dwconv, B, hg, wg)
        tok = conv_bn(tape, z, m.postconv, res=tok)
    outs = []
    for (f, ph, pw), (o, c), m in zip(pyr, offs, net.post_stages):
        xg = _V(tok.t[:, o:o + c])
        xg.pending = []

        def slice_bwd(xg=xg, o=o, c=c):
            if tok.g is None:
                tok.g = torch.empty_like(tok.t)
            dst, acc = tok.g[:, o:o + c], None
            for i, (dz, wt) in enumerate(xg.pending):
                last = i == len(xg.pending) - 1
                acc = ops.gemm_nt(dz, wt, res=acc, out=dst if last else None)
        tape.push(slice_bwd)
        local = conv_bn(tape, f, m.local_embedding)
        ga = conv_bn(tape, xg, m.global_act, defer_dx=True)
        ge = conv_bn(tape, xg, m.global_embedding, defer_dx=True)
        out = _V(ops.inject_fuse(local.t, ga.t, ge.t, B, ph, pw, hg, wg))

        def inj_bwd(out=out, local=local, ga=ga, ge=ge, ph=ph, pw=pw):
            dlocal, dga, dge = ops.inject_bwd(out.g, local.t, ga.t, B, ph, pw, hg, wg)
            tape.acc(local, dlocal), tape.acc(ga, dga), tape.acc(ge, dge)
        tape.push(inj_bwd)
        outs.append((out, ph, pw))
    return outs


def simple_head(tape, model, feats, B, h4, w4):
    """SimpleHead (modules/heads.py:120-170) + classifier weight applied at the head's resolution -> logits4 _V [B*h4*w4, round8(nc)]"""
    T = tape.T
    if (feats[0][1], feats[0][2]) != (h4, w4):
        raise ValueError('TopFormerModel: the first stage must be the finest map')
    if model.fusion == 'sum':
        t = ops.bilinear_fwd(feats[1][0].t, B, feats[1][1], feats[1][2], h4, w4) if len(feats) > 1 else None
        fused = _V(ops.add(feats[0][0].t, t) if t is not None else feats[0][0].t)
        for f, h, w in feats[2:]:
            ops.bilinear_fwd(f.t, B, h, w, h4, w4, out=fused.t, accumulate=True)

        def fuse_bwd():
            tape.acc(feats[0][0], fused.g)
            for f, h, w in feats[1:]:
                tape.acc(f, ops.bilinear_bwd(fused.g, B, h, w, h4, w4))
    else:
        C = feats[0][0].t.shape[1]
        fused = _V(torch.empty((B * h4 * w4, C * len(feats)), dtype=T, device=feats[0][0].t.device))
        for g, (f, h, w) in enumerate(feats):
            ops.bilinear_fwd(f.t, B, h, w, h4, w4, out=fused.t[:, g * C:(g + 1) * C])

        def fuse_bwd():
            for g, (f, h, w) in enumerate(feats):
                tape.acc(f, ops.bilinear_bwd(fused.g[:, g * C:(g + 1) * C], B, h, w, h4, w4))
    tape.push(fuse_bwd)
    hy = conv_bn(tape, fused, model.head.conv)
    cls = model.classifier.net.p_conv.conv
    nc = model.out_channels
    ncp = (nc + 7) // 8 * 8
    wc = torch.zeros((ncp, cls.weight.shape[1]), dtype=F32, device=cls.weight.device)
    wc[:nc] = cls.weight.detach().reshape(nc, -1)
    out = _V(ops.gemm_nt(hy.t, tape.wmat(wc)))

    def cls_bwd():
        tape.pg(cls.weight, ops.gemm_tn(out.g, hy.t)[:nc])
        tape.acc(hy, ops.gemm_nt(out.g, tape.wmat_t(wc)))
    tape.push(cls_bwd)
    return out


def refiner_head(tape, model, feats, B, h4, w4):
    """RefinerHead.forward (modules/heads.py:255-269) + the classifier (core/models.py:103-115) -> logits4 _V [B*h4*w4, round8(nc)]"""
    hd, T = model.head, tape.T
    nc, d = hd.embedding_dim, hd.n_heads
    ncp = (nc + 7) // 8 * 8
    n4 = h4 * w4
    dev = feats[0][0].t.device
    gconv, gn1, gn2 = hd.group_conv.net.p_conv.conv, hd.group_conv.net.norm, hd.norm
    if not isinstance(gn2, torch.nn.GroupNorm):
        raise NotImplementedError('RefinerHead without its final norm is not built on the MI355X path')
    kpar = hd.uni_res.attention.k
    kdev = kpar.detach().float().reshape(1).contiguous()
    wg = gconv.weight.detach().reshape(d, nc, -1)
    u = torch.empty((B * n4, d * ncp), dtype=F32, device=dev)
    saved = []
    for g, (f, h, w) in enumerate(feats):
        mom = ops.image_moments(f.t, B)
        a = ops.simam_apply(f.t, mom, kdev, B)
        wgp = torch.zeros((ncp, wg.shape[2]), dtype=F32, device=dev)
        wgp[:nc] = wg[g]
        s = ops.gemm_nt(a, tape.wmat(wgp), out_f32=True)
        ops.bilinear_fwd(s, B, h, w, h4, w4, out=u[:, g * ncp:(g + 1) * ncp])
        saved.append((mom, a, wgp))
    g1w, g1b, g2w, g2b = (t.detach().float().contiguous() for t in (gn1.weight, gn1.bias, gn2.weight, gn2.bias))
    t1, st1 = ops.groupnorm_fwd(u, B, d, nc, ncp, float(gn1.eps), g1w, g1b, pre_act=ops.SIGMOID)
    # view(n, nc, d, h, w) + Conv3d(d -> 1): r[c] = sum_dd w3[dd] t[c*d + dd]; channel ch = c*d + dd sits in column (ch // nc) * ncp + ch % nc
    w3 = hd.reduce_conv.net.p_conv.conv.weight
    ch = torch.arange(nc * d, device=dev)
    rows_i, cols_i = ch // d, (ch // nc) * ncp + ch % nc
    L1 = torch.zeros((ncp, d * ncp), dtype=F32, device=dev)
    L1[rows_i, cols_i] = w3.detach().float().reshape(d)[ch % d]
    r = ops.gemm_nt(t1, L1)
    if gn2.num_channels != nc or nc % gn2.num_groups:
        raise ValueError('RefinerHead: final GroupNorm does not match the class count')
    cg2 = nc // gn2.num_groups
    t2, st2 = ops.groupnorm_fwd(r, B, gn2.num_groups, cg2, cg2, float(gn2.eps), g2w, g2b)
    cls = model.classifier.net.p_conv.conv
    L2 = torch.zeros((ncp, ncp), dtype=F32, device=dev)
    L2[:nc, :nc] = cls.weight.detach().reshape(nc, nc)
    l4 = ops.gemm_nt(t2, L2)
    out = _V(l4 if T == F32 else ops.cast(l4, T))

    def bwd():
        dl4 = out.g if T == F32 else ops.cast(out.g.contiguous(), F32)
        tape.pg(cls.weight, ops.gemm_tn(dl4, t2)[:nc, :nc].contiguous())
        dt2 = ops.gemm_nt(dl4, L2.t().contiguous())
        dr, dgb2 = ops.groupnorm_bwd(dt2, r, st2, g2w, B, gn2.num_groups, cg2, cg2)
        tape.pg(gn2.weight, dgb2[0]), tape.pg(gn2.bias, dgb2[1])
        dL1 = ops.gemm_tn(dr, t1)
        tape.pg(w3, dL1[rows_i, cols_i].reshape(nc, d).sum(0))
        dt1 = ops.gemm_nt(dr, L1.t().contiguous())
        du, dgb1 = ops.groupnorm_bwd(dt1, u, st1, g1w, B, d, nc, ncp, pre_act=ops.SIGMOID)
        tape.pg(gn1.weight, dgb1[0]), tape.pg(gn1.bias, dgb1[1])
        dk = torch.zeros(1, dtype=F32, device=dev)
        dwg = torch.empty((d, nc, wg.shape[2]), dtype=F32, device=dev)
        for g, ((f, h, w), (mom, a, wgp)) in enumerate(zip(feats, saved)):
            ds = ops.bilinear_bwd(du[:, g * ncp:(g + 1) * ncp], B, h, w, h4, w4)
            if T != F32:
                ds = ops.cast(ds, T)
            dwg[g] = ops.gemm_tn(ds, a)[:nc]
            da = ops.gemm_nt(ds, tape.wmat_t(wgp))
            tape.acc(f, ops.simam_bwd(da, f.t, mom, kdev, B, dk, accumulate_dk=True))
        tape.pg(gconv.weight, dwg)
        tape.pg(kpar, dk)
    tape.push(bwd)
    return out


class _TopFormerTrainFn(GradAwareFunction):
    @staticmethod
    def forward(ctx, model, x, *params):
        T = compute_dtype_for(x)
        tape = _Tape(T, recording())
        B, _, H, W = x.shape
        feats = backbone(tape, model.backbone.net, x)
        h4, w4 = max(f[1] for f in feats), max(f[2] for f in feats)
        if isinstance(model.head, RefinerHead):
            l4 = refiner_head(tape, model, feats, B, h4, w4)
        elif isinstance(model.head, SimpleHead):
            l4 = simple_head(tape, model, feats, B, h4, w4)
        else:
            raise NotImplementedError('TopFormerModel: unknown head')
        full = ops.bilinear_fwd(l4.t, B, h4, w4, H, W)
        ncp = full.shape[1]
        ctx.tape, ctx.l4, ctx.geo, ctx.params = tape, l4, (B, h4, w4, H, W, ncp, model.out_channels), params
        expect_grads(ctx, params, 2)
        return full.view(B, H, W, ncp)[..., :model.out_channels].permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gout):
        tape, l4 = ctx.tape, ctx.l4
        if tape is None or not tape.record:
            raise RuntimeError('TopFormer training step: backward called twice, or the forward ran under torch.no_grad()')
        B, h4, w4, H, W, ncp, nc = ctx.geo
        gp = torch.zeros((B, H, W, ncp), dtype=tape.T, device=gout.device)
        gp[..., :nc].copy_(gout.permute(0, 2, 3, 1))
        l4.g = ops.bilinear_bwd(gp.view(-1, ncp), B, h4, w4, H, W)
        tape.run_backward()
        grads = [tape.pgrads.get(p) if ctx.needs_input_grad[2 + i] else None for i, p in enumerate(ctx.params)]
        ctx.tape = ctx.l4 = None
        return (None, None, *deliver_grads(ctx.params, grads))


def train_forward(model, x):
    """TopFormerModel.forward in train mode -> logits [B, nc, H, W] (a channels-last view), differentiable w.r.t. every parameter."""
    if x.requires_grad:
        raise NotImplementedError('TopFormer training on the MI355X path does not produce a gradient for the input image')
    params = tuple(model.parameters())
    for p in params:
        if p.dtype != F32 or not p.is_contiguous():
            raise TypeError('TopFormer training on the MI355X path keeps fp32, contiguous master parameters')
    return _TopFormerTrainFn.apply(model, x, *params)
