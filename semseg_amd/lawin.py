"""Lawin head on the MI355X kernels.

Same constructors / forward signatures / ``state_dict`` layout as the reference ``modules/lawin.py`` (MLP :7-15,
PatchEmbed :18-35, LawinAttn :39-94, ConvModule :97-105, LawinHead :109-229).  ``LawinHead.forward`` is ONE autograd
Function whose forward and backward are explicit kernel sequences over channels-last activations:

* no unfold: each ratio's context windows are 8x8 gathers from one pooled grid (``ops.lawin_pool_fwd``);
* concatenations are channel slices of pre-allocated buffers that GEMM / BN / bilinear kernels write in place;
* ``cat`` commutes the 1x1 conv with the bilinear x2 up-sampling (both linear): the 5E->E GEMM runs at S/8 and its
  E-channel result is up-sampled, instead of up-sampling five E-channel maps and running the GEMM at S/4
  (4x fewer FLOPs, no [B,5E,S/4,S/4] tensor; exact in real arithmetic, rounding reordered);
* BatchNorm uses batch statistics in training (running buffers updated by the finalize kernel) and running statistics in
  eval; Mish and the Dropout2d multiplier are fused into the BN-apply kernels.
"""
import torch
from torch import nn, Tensor

from . import ops, weightprep
from .runtime import compute_dtype_for, as_tokens_nhwc, deliver_grads, expect_grads, GradAwareFunction
from .segformer import w_fwd, w_bwd, _sub
from .shapes import LAWIN_BRANCHES


def _conv_w(w, dtype):
    """1x1 conv weight [O, I, 1, 1] -> [O, I] in the compute dtype."""
    return w_fwd(w, dtype)


def _rows_on_lines(rows, cols, dtype, dev):
    """[rows, cols] view of a buffer whose row pitch is a multiple of 128 bytes: with E + 48 = 560 bf16 columns a 64-column k-tile of a
    contiguous row straddles two cache lines in 3 rows of 4, and the GEMM tile loaders are bound by the number of lines they request."""
    per = 128 // torch.empty((), dtype=dtype).element_size()
    return torch.empty((rows, (cols + per - 1) // per * per), dtype=dtype, device=dev)[:, :cols]


def _conv_bn_stats(a, w, P, Bf, prefix, training, res=None, out=None):
    """bias-free 1x1 conv (+ res) followed by a BatchNorm's statistics: in training the column sums ride in the GEMM's copy-out"""
    if training:
        return ops.gemm_nt_bn_stats(a, w, res, 1e-5, 0.1, Bf[prefix + '.running_mean'], Bf[prefix + '.running_var'], Bf[prefix + '.num_batches_tracked'],
                                    out=out)
    return ops.gemm_nt(a, w, res=res, out=out), ops.bn_eval_stats(Bf[prefix + '.running_mean'], Bf[prefix + '.running_var'])


def _bn_stats(z, P, Bf, prefix, training):
    if training:
        return ops.bn_stats(z, 1e-5, 0.1, Bf[prefix + '.running_mean'], Bf[prefix + '.running_var'], Bf[prefix + '.num_batches_tracked'])
    return ops.bn_eval_stats(Bf[prefix + '.running_mean'], Bf[prefix + '.running_var'])


def _nchw_view(tok, B, H, W):
    return tok.view(B, H, W, -1).permute(0, 3, 1, 2)


def branch_forward(P, Bf, qtok, G, B, H, W, r, heads, E, training, out=None, fold=False):
    """LawinAttn.forward (modules/lawin.py:57-94) on token matrices.  qtok: the query map [B*H*W, E] in image order (row stride may
    exceed E); G: the pooled context grid of ss_lawin_pool_fwd for ratio r -- for a stand-alone call on 8 x 8 windows (H = W = 8) the
    grid IS the window.  P / Bf: this LawinAttn's parameters / buffers by their own names.  -> (query + BN(conv_out(attention)), state)."""
    T = qtok.dtype
    inter = max(E // 2, 1)
    # operands that the reference builds by looping over / concatenating parameters live in persistent buffers that the
    # batched weight refresh keeps current (weightprep.gather): no torch.stack / torch.cat in the step
    Wm = weightprep.gather([P[f'position_mixing.{i}.weight'] for i in range(heads)], torch.float32).view(heads, 64, 64)
    bm = weightprep.gather([P[f'position_mixing.{i}.bias'] for i in range(heads)], torch.float32).view(heads, 64)
    ctx = ops.lawin_mix(Wm, bm, B, H, W, r, heads, G=G)
    wpg = weightprep.gather([P['phi.weight'], P['g.weight']], T).view(2 * inter, E)
    bpg = weightprep.gather([P['phi.bias'], P['g.bias']], torch.float32)
    pg = ops.gemm_nt(ctx, wpg, bias=bpg)
    th = ops.gemm_nt(qtok, _conv_w(P['theta.weight'], T), bias=P['theta.bias'])
    y, lse = ops.lawin_attn_fwd(th, pg, B, H, W, heads)
    if fold:
        # LawinHead: conv_out's output z goes into the concat slice as it is; the BatchNorm (affine per channel, no activation) and
        # the residual are folded into the weights of the `cat` conv that reads the concat (ops.lawin_cat_fold)
        z, st = _conv_bn_stats(y, _conv_w(P['conv_out.0.weight'], T), P, Bf, 'conv_out.1', training, out=out)
        o = None
    else:
        z, st = _conv_bn_stats(y, _conv_w(P['conv_out.0.weight'], T), P, Bf, 'conv_out.1', training)
        o = ops.bn_apply(z, st, P['conv_out.1.weight'], P['conv_out.1.bias'], act=0, res=qtok, out=out)
    return o, dict(G=G, Wm=Wm, ctx=ctx, pg=pg, th=th, y=y, lse=lse, z=z, st=st)


def branch_backward(P, br, dout, qtok, acc, B, H, W, r, heads, E, training, bn_sums=None):
    """dout: gradient of the branch output [B*H*W, E].  acc: running gradient of the query map (None, or the contributions so far;
    the residual share `dout` itself is NOT added here).  -> (acc + theta path, dG fp32 [grid rows, E], {param: grad})."""
    T = dout.dtype
    inter, G = max(E // 2, 1), {}
    if bn_sums is None:
        dz, dgb = ops.bn_bwd(dout, br['z'], br['st'], P['conv_out.1.weight'], P['conv_out.1.bias'], act=0, training=training)
    else:
        # LawinHead: the first-pass sums came out of the cat conv's weight gradient (ops.lawin_branch_bn_sums): second pass only
        dgb, coef = bn_sums
        dz = ops.bn_bwd_coef(dout, br['z'], br['st'], P['conv_out.1.weight'], P['conv_out.1.bias'], coef)
    G['conv_out.1.weight'], G['conv_out.1.bias'] = dgb[0], dgb[1]
    G['conv_out.0.weight'] = ops.gemm_tn(dz, br['y'])
    dy = ops.gemm_nt(dz, w_bwd(P['conv_out.0.weight'], T))
    dth, dpg = ops.lawin_attn_bwd(br['th'], br['pg'], br['y'], dy, br['lse'], B, H, W, heads)
    G['theta.weight'], G['theta.bias'] = ops.gemm_tn(dth, qtok, colsum=True)
    acc = ops.gemm_nt(dth, w_bwd(P['theta.weight'], T), res=acc)
    dwpg, dbpg = ops.gemm_tn(dpg, br['ctx'], colsum=True)
    G['phi.weight'], G['g.weight'] = dwpg[:inter], dwpg[inter:]
    G['phi.bias'], G['g.bias'] = dbpg[:inter], dbpg[inter:]
    wpg32 = weightprep.gather([P['phi.weight'], P['g.weight']], torch.float32).view(2 * inter, E)
    dctxp = ops.gemm_nt(dpg, ops.transpose_cast(wpg32, T))
    dWb = ops.lawin_mix_wgrad(br['G'], dctxp, B, H, W, r, heads)
    for i in range(heads):
        G[f'position_mixing.{i}.weight'] = dWb[i, :, :64]
        G[f'position_mixing.{i}.bias'] = dWb[i, :, 64]
    dctx = ops.lawin_mix(br['Wm'], None, B, H, W, r, heads, xwin=dctxp, transpose=True)
    return acc, ops.lawin_gather_bwd(dctx, B, H, W, r), G


def _cat_fold_operands(P, brs):
    """(stats, gammas, betas) of the three branch BatchNorms in concat order (r = 8, 4, 2)"""
    names = [f'lawin_{r}.conv_out.1' for r, _ in LAWIN_BRANCHES]
    return [br['st'] for br in brs], [P[n + '.weight'].detach() for n in names], [P[n + '.bias'].detach() for n in names]


def head_forward(P, Bf, feats, E, training, drop_mask):
    """feats: 4 NCHW-shaped maps (S/4..S/32).  -> logits tokens [B*H4*W4, nc], saved state."""
    T = compute_dtype_for(feats[0])
    B = feats[0].shape[0]
    dims = [f.shape[1] for f in feats]
    hw = [(f.shape[2], f.shape[3]) for f in feats]
    (H4, W4), (H8, W8) = hw[0], hw[1]
    if H8 % 8 or W8 % 8:
        raise ValueError(f'LawinHead: the S/8 map must be a multiple of 8 (got {H8}x{W8}); input tiles must be multiples of 64')
    t = [as_tokens_nhwc(f.detach(), T) for f in feats]
    R8, R4, inter = B * H8 * W8, B * H4 * W4, max(E // 2, 1)
    dev = t[0].device
    S = {'t': t, 'hw': hw, 'dims': dims, 'T': T, 'B': B}

    # per-level linear embeddings; linear_fuse's 1x1 conv over their up-sampled concat [c4, c3, c2] commutes with the resize (both
    # linear): each level's E x E block of the conv runs at the level's own resolution and the E-channel results are up-sampled
    # and summed -- no [R8, 3E] concat, 1/16 and 1/4 of the conv's work for c4 and c3 (exact in real arithmetic, rounding reordered)
    Wf = _conv_w(P['linear_fuse.conv.weight'], T)
    cl = [None] + [ops.gemm_nt(t[lvl], w_fwd(P[f'linear_c{lvl + 1}.proj.weight'], T), bias=P[f'linear_c{lvl + 1}.proj.bias']) for lvl in (1, 2, 3)]
    g2, g3 = ops.gemm_nt(cl[2], Wf[:, E:2 * E]), ops.gemm_nt(cl[3], Wf[:, :E])
    if E % 8 == 0:
        U = ops.bilinear_fwd2(g2, hw[2][0], hw[2][1], g3, hw[3][0], hw[3][1], B, H8, W8)       # both levels in one pass over U
    else:
        U = ops.bilinear_fwd(g2, B, hw[2][0], hw[2][1], H8, W8)
        U = ops.bilinear_fwd(g3, B, hw[3][0], hw[3][1], H8, W8, out=U, accumulate=True)
    zf, stf = _conv_bn_stats(cl[1], Wf[:, 2 * E:], P, Bf, 'linear_fuse.bn', training, res=U)
    cat5 = torch.empty((R8, 4 * E), dtype=T, device=dev)          # [feat, z_8, z_4, z_2] (conv_out outputs); the pooled branch is a vector per image
    feat = ops.bn_apply(zf, stf, P['linear_fuse.bn.weight'], P['linear_fuse.bn.bias'], act=1, out=cat5[:, :E])
    S.update(cl=cl, zf=zf, stf=stf, cat5=cat5)

    # three large-window attention branches
    S['br'] = []
    for k, (r, heads) in enumerate(LAWIN_BRANCHES):
        G, amax = ops.lawin_pool_fwd(feat, B, H8, W8, r)
        _, br = branch_forward(_sub(P, f'lawin_{r}.'), _sub(Bf, f'lawin_{r}.'), feat, G, B, H8, W8, r, heads, E, training,
                               out=cat5[:, (k + 1) * E:(k + 2) * E], fold=True)
        br['amax'] = amax
        S['br'].append(br)

    # image-pool branch: GAP -> ConvModule -> broadcast (bilinear of a 1x1 map)
    gp = ops.cast(ops.colsum(feat, groups=B, scale=1.0 / (H8 * W8)), T)
    zp = ops.gemm_nt(gp, _conv_w(P['image_pool.1.conv.weight'], T))
    stp = _bn_stats(zp, P, Bf, 'image_pool.1.bn', training)
    ap = ops.bn_apply(zp, stp, P['image_pool.1.bn.weight'], P['image_pool.1.bn.bias'], act=1)
    S.update(gp=gp, zp=zp, stp=stp, ap=ap)

    # cat: 1x1 conv at S/8, bilinear x2, BN + Mish at S/4; low-level fuse; classifier
    # (the S/4 map z4 = resize(z8) is never written: the BatchNorm kernels read it through its bilinear taps, ss_bn_*_up)
    # The image-pool branch is constant over an image, so its E input columns of the conv are one vector per image (vb) that
    # commutes with the resize and joins the map inside the BatchNorm kernels: the conv runs on 4E columns, not 5E.
    Wc = _conv_w(P['cat.conv.weight'], T)
    fold = _cat_fold_operands(P, S['br'])
    Wcf, Wct_sum, cconst = ops.lawin_cat_fold(P['cat.conv.weight'].detach().view(E, 5 * E), *fold, E, T)
    vb = ops.gemm_nt(ap, Wc[:, 4 * E:], bias=cconst, out_f32=True)
    z8 = ops.gemm_nt(cat5, Wcf)
    if training:
        stc = ops.bn_stats_up(z8, B, H8, W8, H4, W4, 1e-5, 0.1, Bf['cat.bn.running_mean'], Bf['cat.bn.running_var'],
                              Bf['cat.bn.num_batches_tracked'], bvec=vb)
    else:
        stc = ops.bn_eval_stats(Bf['cat.bn.running_mean'], Bf['cat.bn.running_var'])
    # the concat [up(cat) | linear_c1] has E + 48 columns: its rows are padded to whole 64-column k-tiles (128-byte lines) with columns
    # that are written as exact zeros by linear_c1 itself (zero weight rows / bias entries behind the real ones) and meet zero weight
    # columns in low_level_fuse, so that GEMM runs on the 256-row tile kernels (K % 64 == 0)
    c1 = P['linear_c1.proj.weight'].shape[0]
    Kc = (E + c1 + 63) // 64 * 64
    cat2f = torch.empty((R4, Kc), dtype=T, device=dev)
    cat2 = cat2f[:, :E + c1]
    ops.bn_apply_up(z8, B, H8, W8, H4, W4, stc, P['cat.bn.weight'], P['cat.bn.bias'], act=1, out=cat2[:, :E], bvec=vb)
    ops.gemm_nt(t[0], weightprep.cast_padded(P['linear_c1.proj.weight'], T, Kc - E), bias=weightprep.cast_padded(P['linear_c1.proj.bias'], torch.float32, Kc - E).view(-1),
                out=cat2f[:, E:])
    zl, stl = _conv_bn_stats(cat2f, weightprep.pack(P['low_level_fuse.conv.weight'], T, k_multiple=64), P, Bf, 'low_level_fuse.bn', training)
    nc = P['linear_pred.weight'].shape[0]
    rpi4 = H4 * W4 if drop_mask is not None else 0
    if ops.bn_classify_ok(E, nc):
        # BatchNorm + Mish + Dropout2d + classifier in one pass: the E-channel activation at S/4 is never written
        fused = None
        seg = ops.bn_apply_classify(zl, stl, P['low_level_fuse.bn.weight'], P['low_level_fuse.bn.bias'],
                                    P['linear_pred.weight'].detach().view(nc, E), P['linear_pred.bias'].detach(), mask=drop_mask, rows_per_image=rpi4)
    else:
        fused = ops.bn_apply(zl, stl, P['low_level_fuse.bn.weight'], P['low_level_fuse.bn.bias'], act=1, mask=drop_mask, rows_per_image=rpi4)
        seg = ops.gemm_nt(fused, _conv_w(P['linear_pred.weight'], T), bias=P['linear_pred.bias'])
    S.update(z8=z8, vb=vb, Wct_sum=Wct_sum, stc=stc, cat2=cat2, zl=zl, stl=stl, fused=fused, drop_mask=drop_mask, training=training)
    return seg, S


def head_backward(P, S, dseg, E):
    """dseg: [B*H4*W4, nc] -> (grads of the 4 feature token matrices, {param: grad})."""
    T, B, t, hw = S['T'], S['B'], S['t'], S['hw']
    (H4, W4), (H8, W8) = hw[0], hw[1]
    inter, training, G = max(E // 2, 1), S['training'], {}
    mask = S['drop_mask']
    rpi4 = H4 * W4 if mask is not None else 0

    def bn_grads(prefix, dgb):
        G[prefix + '.weight'], G[prefix + '.bias'] = dgb[0], dgb[1]

    # classifier + low-level fuse
    G['linear_pred.bias'] = ops.colsum_any(dseg)
    if S['fused'] is None:
        nc = P['linear_pred.weight'].shape[0]
        dzl, dgb, G['linear_pred.weight'] = ops.bn_bwd_classify(dseg, S['zl'], S['stl'], P['low_level_fuse.bn.weight'], P['low_level_fuse.bn.bias'],
                                                                P['linear_pred.weight'].detach().view(nc, E), mask=mask, rows_per_image=rpi4,
                                                                training=training)
    else:
        G['linear_pred.weight'] = ops.gemm_tn(dseg, S['fused'])
        dfused = ops.gemm_nt(dseg, w_bwd(P['linear_pred.weight'], T))
        dzl, dgb = ops.bn_bwd(dfused, S['zl'], S['stl'], P['low_level_fuse.bn.weight'], P['low_level_fuse.bn.bias'], act=1, mask=mask,
                              rows_per_image=rpi4, training=training)
    bn_grads('low_level_fuse.bn', dgb)
    Kc2 = S['cat2'].shape[1]
    if T == torch.bfloat16 and Kc2 % 8 == 0 and 320 < Kc2 <= 576 and E % 64 == 0 and S['cat2'].stride(0) == 576:
        # E + 48 = 560 columns suit neither the 256- nor the 320-column tile kernels (they ran on 128-column tiles at 0.56 PFLOP/s):
        # both gradients run as a 320-column product plus a 256-column one over the concat padded to 576 columns (zero weight rows
        # / the concat's own zero columns behind column 560); dzl is read twice, the products run on the wide tiles
        wt = weightprep.transpose_padded(P['low_level_fuse.conv.weight'], E, Kc2, T, 576)          # [576, E]
        dfull = torch.empty((dzl.shape[0], 576), dtype=T, device=dzl.device)
        ops.gemm_nt(dzl, wt[:320], out=dfull[:, :320])
        ops.gemm_nt(dzl, wt[320:], out=dfull[:, 320:])
        dcat2 = dfull[:, :Kc2]
        cat2f = S['cat2'].as_strided((S['cat2'].shape[0], 576), (576, 1))                        # the padded buffer the forward filled
        Gl = torch.empty((E, 576), dtype=torch.float32, device=dzl.device)
        ops.gemm_tn(dzl, cat2f[:, :320], out=Gl[:, :320])
        ops.gemm_tn(dzl, cat2f[:, 320:], out=Gl[:, 320:])
        G['low_level_fuse.conv.weight'] = Gl[:, :Kc2]                                             # row-strided: deliver_grads takes it as it is
    else:
        G['low_level_fuse.conv.weight'] = ops.gemm_tn(dzl, S['cat2'])
        dcat2 = ops.gemm_nt(dzl, w_bwd(P['low_level_fuse.conv.weight'], T), out=_rows_on_lines(dzl.shape[0], Kc2, T, dzl.device))
    dc1 = dcat2[:, E:]
    G['linear_c1.proj.weight'], G['linear_c1.proj.bias'] = ops.gemm_tn(dc1, t[0], colsum=True)
    dt = [ops.gemm_nt(dc1, w_bwd(P['linear_c1.proj.weight'], T)), None, None, None]
    # cat
    dz8, dgb = ops.bn_bwd_up_to_input(dcat2[:, :E], S['z8'], B, H8, W8, H4, W4, S['stc'], P['cat.bn.weight'], P['cat.bn.bias'], act=1,
                                      training=training, bvec=S['vb'], overwrite_dy=True)
    bn_grads('cat.bn', dgb)
    Gc = torch.empty((E, 5 * E), dtype=torch.float32, device=dz8.device)
    G['cat.conv.weight'] = Gc
    ops.gemm_tn(dz8, S['cat5'], out=Gc[:, :4 * E])
    Wct = w_bwd(P['cat.conv.weight'], T)                        # [5E, E]
    # columns E..4E: gradients of the concat's slices feat + BN(z_r); columns 0..E: the SUM of all four slice gradients, i.e. the fan-in
    # on feat (short path + three residuals) straight from the GEMM over the summed weight block
    dcat5 = ops.gemm_nt(dz8, S['Wct_sum'])
    feat = S['cat5'][:, :E]
    # image-pool branch: the gradient of the per-image vector is the per-image column sum of dz8 (the resize preserves sums)
    dvb32 = ops.colsum(dz8, groups=B)
    fold = _cat_fold_operands(P, S['br'])
    br_dgb, br_coef = ops.lawin_branch_bn_sums(Gc, P['cat.conv.weight'].detach().view(E, 5 * E), fold[0], dvb32, B, E, B * H8 * W8, training)
    ops.lawin_cat_wgrad_fix(Gc, *fold, dvb32, B, E)
    dvb = ops.cast(dvb32, T)
    ops.gemm_tn(dvb, S['ap'], out=Gc[:, 4 * E:])
    dap = ops.gemm_nt(dvb, Wct[4 * E:])
    dzp, dgb = ops.bn_bwd(dap, S['zp'], S['stp'], P['image_pool.1.bn.weight'], P['image_pool.1.bn.bias'], act=1, training=training)
    bn_grads('image_pool.1.bn', dgb)
    G['image_pool.1.conv.weight'] = ops.gemm_tn(dzp, S['gp'])
    # gradient fan-in on feat: short path + the three residuals + broadcast of the pooled branch, whose 1/(H8*W8) rides in the
    # GEMM epilogue as a row scale
    pool_vec = ops.gemm_nt(dzp, w_bwd(P['image_pool.1.conv.weight'], T), out_f32=True, rowscale=_const(dzp.device, 1.0 / (H8 * W8)),
                           rows_per_scale=dzp.shape[0])
    acc = dcat5[:, :E]                                          # pool_vec (the broadcast of the pooled branch) joins in lawin_pool_bwd
    dGs, amaxs = [], []
    for k, (r, heads) in enumerate(LAWIN_BRANCHES):
        p, br = f'lawin_{r}', S['br'][k]
        acc, dG, Gb = branch_backward(_sub(P, p + '.'), br, dcat5[:, (k + 1) * E:(k + 2) * E], feat, acc, B, H8, W8, r, heads, E, training,
                                      bn_sums=(br_dgb[k], br_coef[k]))
        G.update({f'{p}.{n}': g for n, g in Gb.items()})
        dGs.append(dG)
        amaxs.append(br['amax'])
    dfeat = ops.lawin_pool_bwd(dGs, amaxs, B, H8, W8, E, T, base=acc, bvec=pool_vec)
    # linear_fuse and the per-level embeddings
    dzf, dgb = ops.bn_bwd(dfeat, S['zf'], S['stf'], P['linear_fuse.bn.weight'], P['linear_fuse.bn.bias'], act=1, training=training)
    bn_grads('linear_fuse.bn', dgb)
    cl = S['cl']
    Gf = torch.empty((E, 3 * E), dtype=torch.float32, device=dzf.device)
    G['linear_fuse.conv.weight'] = Gf
    Wft = w_bwd(P['linear_fuse.conv.weight'], T)               # [3E, E]
    for lvl, sl in ((1, slice(2 * E, 3 * E)), (2, slice(E, 2 * E)), (3, slice(0, E))):
        dg = dzf if lvl == 1 else ops.bilinear_bwd(dzf, B, hw[lvl][0], hw[lvl][1], H8, W8)      # gradient of the level's conv output
        ops.gemm_tn(dg, cl[lvl], out=Gf[:, sl])
        d = ops.gemm_nt(dg, Wft[sl])
        name = f'linear_c{lvl + 1}.proj'
        G[name + '.weight'], G[name + '.bias'] = ops.gemm_tn(d, t[lvl], colsum=True)
        dt[lvl] = ops.gemm_nt(d, w_bwd(P[name + '.weight'], T))
    return dt, G


_CONSTS = {}


def _const(device, value):
    """one-element fp32 device tensor holding `value` (made once per device and value)"""
    key = (device, float(value))
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.full((1,), float(value), dtype=torch.float32, device=device)
    return t


class LawinHeadFn(GradAwareFunction):
    @staticmethod
    def forward(ctx, module, drop_mask, names, x1, x2, x3, x4, *params):
        P = dict(zip(names, params))
        Bf = dict(module.named_buffers())
        seg, S = head_forward(P, Bf, (x1, x2, x3, x4), module.embed_dim, module.training, drop_mask)
        ctx.S, ctx.names, ctx.params, ctx.E = S, names, params, module.embed_dim
        expect_grads(ctx, params, 7)
        B, (H4, W4) = S['B'], S['hw'][0]
        return _nchw_view(seg, B, H4, W4)

    @staticmethod
    def backward(ctx, dout):
        S = ctx.S
        P = dict(zip(ctx.names, ctx.params))
        dseg = as_tokens_nhwc(dout, S['T'])
        dt, G = head_backward(P, S, dseg.contiguous(), ctx.E)
        ctx.S = None
        B, hw = S['B'], S['hw']
        dfe = [(_nchw_view(d, B, h, w) if ctx.needs_input_grad[3 + i] else None) for i, (d, (h, w)) in enumerate(zip(dt, hw))]
        grads = [G.get(n) for n in ctx.names]
        grads = deliver_grads(ctx.params, [g if g is None else g.reshape(p.shape) for g, p in zip(grads, ctx.params)])
        return (None, None, None, *dfe, *grads)


class _LawinFeatsFn(GradAwareFunction):
    @staticmethod
    def forward(ctx, module, names, x, *params):
        P, Bf = dict(zip(names, params)), dict(module.named_buffers())
        T, E = compute_dtype_for(x), module.embed_dim
        B, _, H, W = x.shape
        feat = as_tokens_nhwc(x.detach(), T)
        outs, brs = [], []
        for r, heads in LAWIN_BRANCHES:
            G, amax = ops.lawin_pool_fwd(feat, B, H, W, r)
            o, br = branch_forward(_sub(P, f'lawin_{r}.'), _sub(Bf, f'lawin_{r}.'), feat, G, B, H, W, r, heads, E, module.training)
            br['amax'] = amax
            outs.append(_nchw_view(o, B, H, W))
            brs.append(br)
        ctx.S, ctx.names, ctx.params, ctx.mod, ctx.geom = (feat, brs), names, params, module, (B, H, W)
        expect_grads(ctx, params, 3)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        (feat, brs), module, (B, H, W) = ctx.S, ctx.mod, ctx.geom
        P, E, T = dict(zip(ctx.names, ctx.params)), module.embed_dim, feat.dtype
        d = [as_tokens_nhwc(g, T).contiguous() for g in douts]
        acc = d[0] + d[1] + d[2]                                   # the three residual paths
        G, dGs = {}, []
        for k, (r, heads) in enumerate(LAWIN_BRANCHES):
            acc, dG, Gb = branch_backward(_sub(P, f'lawin_{r}.'), brs[k], d[k], feat, acc, B, H, W, r, heads, E, module.training)
            G.update({f'lawin_{r}.{n}': g for n, g in Gb.items()})
            dGs.append(dG)
        dfeat = ops.lawin_pool_bwd(dGs, [br['amax'] for br in brs], B, H, W, E, T, base=acc)
        ctx.S = None
        grads = [G.get(n) for n in ctx.names]
        grads = deliver_grads(ctx.params, [g if g is None else g.reshape(p.shape) for g, p in zip(grads, ctx.params)])
        return (None, None, _nchw_view(dfeat, B, H, W) if ctx.needs_input_grad[2] else None, *grads)


# ------------------------------------------------------------------ modules (reference API)
class MLP(nn.Module):
    def __init__(self, dim=2048, embed_dim=768):
        super().__init__()
        self.proj = nn.Linear(dim, embed_dim)

    def forward(self, x: Tensor) -> Tensor:
        from .segformer import _ParamFn

        class _Fn(_ParamFn):
            @staticmethod
            def fwd(P, t, meta):
                return ops.gemm_nt(t, w_fwd(P['proj.weight'], t.dtype), bias=P['proj.bias']), {'t': t}

            @staticmethod
            def bwd(P, S, dy, meta):
                return ops.gemm_nt(dy, w_bwd(P['proj.weight'], dy.dtype)), {'proj.weight': ops.gemm_tn(dy, S['t']), 'proj.bias': ops.colsum_any(dy)}

        b, c, h, w = x.shape
        return _Fn.run(as_tokens_nhwc(x, compute_dtype_for(x)), None, self).view(b, h * w, -1)


class _PatchPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ps):
        b, c, h, w = x.shape
        T = compute_dtype_for(x)
        y, amax = ops.patch_pool_fwd(as_tokens_nhwc(x.detach(), T), b, h, w, ps)
        ctx.amax, ctx.geom, ctx.T = amax, (b, h, w, ps), T
        return _nchw_view(y, b, -(-h // ps), -(-w // ps))

    @staticmethod
    def backward(ctx, dy):
        b, h, w, ps = ctx.geom
        dx = ops.patch_pool_bwd(as_tokens_nhwc(dy, ctx.T).contiguous(), ctx.amax, b, h, w, ps)
        return _nchw_view(dx, b, h, w), None


class PatchEmbed(nn.Module):
    """0.5 * (max_pool + avg_pool) with right/bottom zero padding (modules/lawin.py:18-35) on ss_patch_pool_fwd / _bwd.
    LawinHead.forward itself never calls it: its pooled-context grid (ss_lawin_pool_fwd) subsumes the three ds_r modules."""

    def __init__(self, patch_size=4) -> None:
        super().__init__()
        self.patch_size = patch_size
        self.proj = nn.ModuleList([nn.MaxPool2d(patch_size, patch_size), nn.AvgPool2d(patch_size, patch_size)])

    def forward(self, x: Tensor) -> Tensor:
        if x.shape[1] % 8:
            raise ValueError('PatchEmbed: channels must be a multiple of 8 on the MI355X path')
        return _PatchPoolFn.apply(x, self.patch_size)


class _LawinAttnFn(GradAwareFunction):
    @staticmethod
    def forward(ctx, module, names, query, context, *params):
        P, Bf = dict(zip(names, params)), dict(module.named_buffers())
        T = compute_dtype_for(query)
        bw, E = query.shape[0], query.shape[1]
        q, c = as_tokens_nhwc(query.detach(), T), as_tokens_nhwc(context.detach(), T).contiguous()
        # each 8 x 8 window is an "image" of one window: its pooled grid is the context window itself (any ratio)
        o, br = branch_forward(P, Bf, q, c, bw, 8, 8, 8, module.head, E, module.training)
        ctx.S, ctx.names, ctx.params, ctx.mod = (q, br), names, params, module
        expect_grads(ctx, params, 4)
        return _nchw_view(o, bw, 8, 8)

    @staticmethod
    def backward(ctx, dout):
        (q, br), module = ctx.S, ctx.mod
        P = dict(zip(ctx.names, ctx.params))
        bw, E, T = q.shape[0] // 64, q.shape[1], q.dtype
        d = as_tokens_nhwc(dout, T).contiguous()
        dq, dG, G = branch_backward(P, br, d, q, d, bw, 8, 8, 8, module.head, E, module.training)
        ctx.S = None
        grads = [G.get(n) for n in ctx.names]
        grads = deliver_grads(ctx.params, [g if g is None else g.reshape(p.shape) for g, p in zip(grads, ctx.params)])
        return (None, None, _nchw_view(dq, bw, 8, 8) if ctx.needs_input_grad[2] else None,
                _nchw_view(ops.cast(dG, T), bw, 8, 8) if ctx.needs_input_grad[3] else None, *grads)


class LawinAttn(nn.Module):
    """modules/lawin.py:39-94: per-head position mixing of the context, theta / phi / g 1x1 convs, per-head softmax attention over the
    64 positions, conv_out (1x1 + BatchNorm2d), + query.  Inside LawinHead.forward the same kernel sequence runs on the whole map
    (branch_forward); forward() here is the reference's stand-alone call on window batches [windows, C, 8, 8]."""

    def __init__(self, in_ch=512, head=4, patch_size=8, reduction=2) -> None:
        super().__init__()
        self.head = head
        self.position_mixing = nn.ModuleList([nn.Linear(patch_size * patch_size, patch_size * patch_size) for _ in range(self.head)])
        self.inter_channels = max(in_ch // reduction, 1)
        self.g = nn.Conv2d(in_ch, self.inter_channels, (1, 1))
        self.theta = nn.Conv2d(in_ch, self.inter_channels, (1, 1))
        self.phi = nn.Conv2d(in_ch, self.inter_channels, (1, 1))
        self.conv_out = nn.Sequential(nn.Conv2d(self.inter_channels, in_ch, (1, 1), bias=False), nn.BatchNorm2d(in_ch))
        self.patch_size, self.in_ch = patch_size, in_ch

    def forward(self, query: Tensor, context: Tensor) -> Tensor:
        if query.shape != context.shape or tuple(query.shape[1:]) != (self.in_ch, 8, 8) or self.patch_size != 8:
            raise ValueError(f'LawinAttn: query and context must both be [windows, {self.in_ch}, 8, 8] (position mixing is Linear(64, 64)); '
                             f'got {tuple(query.shape)} and {tuple(context.shape)}')
        if self.in_ch % (2 * self.head) or (self.in_ch // 2) % 8:
            raise ValueError('LawinAttn: in_ch / 2 must be a multiple of 8 and of the head count on the MI355X path')
        names = [n for n, _ in self.named_parameters()]
        return _LawinAttnFn.apply(self, names, query, context, *[p for _, p in self.named_parameters()])


class ConvModule(nn.Module):
    """1x1 conv (no bias) -> BatchNorm2d -> Mish (modules/lawin.py:97-105)."""

    def __init__(self, c1, c2):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, (1, 1), bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.activate = nn.Mish(True)

    def forward(self, x: Tensor) -> Tensor:
        from .segformer import _ParamFn
        mod = self

        class _Fn(_ParamFn):
            @staticmethod
            def fwd(P, t, meta):
                z = ops.gemm_nt(t, _conv_w(P['conv.weight'], t.dtype))
                st = _bn_stats(z, P, dict(mod.named_buffers()), 'bn', mod.training)
                return ops.bn_apply(z, st, P['bn.weight'], P['bn.bias'], act=1), {'t': t, 'z': z, 'st': st}

            @staticmethod
            def bwd(P, S, dy, meta):
                dz, dgb = ops.bn_bwd(dy, S['z'], S['st'], P['bn.weight'], P['bn.bias'], act=1, training=mod.training)
                return ops.gemm_nt(dz, w_bwd(P['conv.weight'], dy.dtype)), {'conv.weight': ops.gemm_tn(dz, S['t']), 'bn.weight': dgb[0], 'bn.bias': dgb[1]}

        b, c, h, w = x.shape
        y = _Fn.run(as_tokens_nhwc(x, compute_dtype_for(x)), None, self)
        return _nchw_view(y, b, h, w)


class LawinHead(nn.Module):
    def __init__(self, in_channels: list, embed_dim=512, out_channels=8) -> None:
        super().__init__()
        if embed_dim % 128:
            raise ValueError('LawinHead: embed_dim must be a multiple of 128 (64 heads on embed_dim/2 channels, modules/lawin.py:48,73,122)')
        for i, dim in enumerate(in_channels):
            self.add_module(f"linear_c{i + 1}", MLP(dim, 48 if i == 0 else embed_dim))
        self.linear_fuse = ConvModule(embed_dim * 3, embed_dim)
        self.lawin_8 = LawinAttn(embed_dim, 64)
        self.lawin_4 = LawinAttn(embed_dim, 16)
        self.lawin_2 = LawinAttn(embed_dim, 4)
        self.ds_8, self.ds_4, self.ds_2 = PatchEmbed(8), PatchEmbed(4), PatchEmbed(2)
        self.image_pool = nn.Sequential(nn.AdaptiveAvgPool2d(1), ConvModule(embed_dim, embed_dim))
        self.cat = ConvModule(embed_dim * 5, embed_dim)
        self.low_level_fuse = ConvModule(embed_dim + 48, embed_dim)
        self.linear_pred = nn.Conv2d(embed_dim, out_channels, (1, 1))
        self.dropout = nn.Dropout2d(0.1)
        self.embed_dim, self.out_channels = embed_dim, out_channels

    def _drop_mask(self, b, dtype, device):
        p = self.dropout.p
        if not self.training or p == 0.0:
            return None
        noise = torch.empty((b, self.embed_dim, 1, 1), dtype=dtype, device=device).bernoulli_(1 - p).div_(1 - p)   # as F.dropout2d
        return noise.float().reshape(b, self.embed_dim).contiguous()

    def get_lawin_att_feats(self, x: Tensor, patch_size: int):
        """modules/lawin.py:140-179: the three large-window attention maps [lawin_8, lawin_4, lawin_2](x), each [B, C, H, W].  The
        unfold / pool / fold of the reference is the pooled-grid gather of ss_lawin_pool_fwd; only patch_size 8 exists (position
        mixing is Linear(64, 64) in the reference too)."""
        if patch_size != 8:
            raise ValueError('LawinHead.get_lawin_att_feats: patch_size must be 8 (LawinAttn.position_mixing is Linear(64, 64))')
        _, c, h, w = x.shape
        if c != self.embed_dim or h % 8 or w % 8:
            raise ValueError(f'LawinHead.get_lawin_att_feats: need [B, {self.embed_dim}, 8k, 8l], got {tuple(x.shape)}')
        named = [(n, p) for n, p in self.named_parameters() if n.startswith('lawin_')]
        return list(_LawinFeatsFn.apply(self, [n for n, _ in named], x, *[p for _, p in named]))

    def forward(self, features):
        x1, x2, x3, x4 = features
        T = compute_dtype_for(x1)
        if self.training and x1.shape[0] < 2:
            raise ValueError('Expected more than 1 value per channel when training (image_pool BatchNorm on a (B,C,1,1) map needs B >= 2)')
        names = [n for n, _ in self.named_parameters()]
        params = [p for _, p in self.named_parameters()]
        return LawinHeadFn.apply(self, self._drop_mask(x1.shape[0], T, x1.device), names, x1, x2, x3, x4, *params)
