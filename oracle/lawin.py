"""CPU oracle: Lawin head.  TEST INFRASTRUCTURE ONLY.

Functional restatement of /root/reference/modules/lawin.py on a flat
``state_dict`` with the reference ``LawinHead.state_dict()`` keys.  It follows
the reference's own (unfold -> pool) formulation, NOT the pooled-grid
restructuring the HIP path uses, so it is an independent check of it.
"""
import torch
import torch.nn.functional as F

BRANCHES = ((8, 64), (4, 16), (2, 4))   # (r, heads)  modules/lawin.py:122-124


def _bn(sd, p, x, training, momentum=0.1, eps=1e-5):
    """nn.BatchNorm2d; running stats in ``sd`` are updated in place when training."""
    if training and p + '.num_batches_tracked' in sd:
        sd[p + '.num_batches_tracked'] += 1
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'],
                        sd[p + '.weight'], sd[p + '.bias'], training, momentum, eps)


def conv_module(sd, p, x, training):
    """modules/lawin.py:97-105: 1x1 conv (no bias) -> BN -> Mish."""
    return F.mish(_bn(sd, p + '.bn', F.conv2d(x, sd[p + '.conv.weight']), training))


def level_mlp(sd, p, x):
    """modules/lawin.py:7-15 followed by the permute/reshape of :184-186."""
    b, _, h, w = x.shape
    y = F.linear(x.flatten(2).transpose(1, 2), sd[p + '.proj.weight'], sd[p + '.proj.bias'])
    return y.permute(0, 2, 1).reshape(b, -1, h, w)


def windows(x, size, stride, pad):
    """F.unfold + rearrange of modules/lawin.py:142-167 ->  (b*nh*nw, c, size, size)."""
    b, c, _, _ = x.shape
    xp = F.pad(x, (pad, pad, pad, pad))
    win = xp.unfold(2, size, stride).unfold(3, size, stride)   # b,c,nh,nw,size,size
    nh, nw = win.shape[2:4]
    return win.permute(0, 2, 3, 1, 4, 5).reshape(b * nh * nw, c, size, size), nh, nw


def pooled(x, r):
    """modules/lawin.py:27-35 (inputs here are always divisible by r)."""
    return 0.5 * (F.max_pool2d(x, r, r) + F.avg_pool2d(x, r, r))


def lawin_attn(sd, p, query, context, head, training):
    """modules/lawin.py:57-94."""
    b, c, h, w = context.shape
    ctx = context.reshape(b, c, -1)
    cg = c // head
    mixed = [F.linear(ctx[:, cg * i:cg * (i + 1), :],
                      sd[f'{p}.position_mixing.{i}.weight'],
                      sd[f'{p}.position_mixing.{i}.bias']) for i in range(head)]
    ctx = (ctx + torch.cat(mixed, dim=1)).reshape(b, c, h, w)
    inter = sd[p + '.g.weight'].shape[0]
    d = inter // head

    def split(t):  # "b (h dim) n -> (b h) dim n"
        return t.reshape(b, head, d, -1).reshape(b * head, d, -1)

    g = split(F.conv2d(ctx, sd[p + '.g.weight'], sd[p + '.g.bias'])).transpose(1, 2)
    th = split(F.conv2d(query, sd[p + '.theta.weight'], sd[p + '.theta.bias'])).transpose(1, 2)
    ph = split(F.conv2d(ctx, sd[p + '.phi.weight'], sd[p + '.phi.bias']))
    pw = torch.softmax(torch.matmul(th, ph) / (d ** 0.5), dim=-1)
    y = torch.matmul(pw, g)                                   # (b h) n dim
    y = y.reshape(b, head, -1, d).permute(0, 1, 3, 2).reshape(b, inter, *query.shape[-2:])
    z = F.conv2d(y, sd[p + '.conv_out.0.weight'])
    return query + _bn(sd, p + '.conv_out.1', z, training)


def lawin_branches(sd, feat, training, patch=8):
    """modules/lawin.py:140-179."""
    b, c, h, w = feat.shape
    query, nh, nw = windows(feat, patch, patch, 0)
    outs = []
    for r, head in BRANCHES:
        ctx, _, _ = windows(feat, patch * r, patch, int((r - 1) / 2 * patch))
        o = lawin_attn(sd, f'lawin_{r}', query, pooled(ctx, r), head, training)
        o = o.reshape(b, nh, nw, c, patch, patch).permute(0, 3, 1, 4, 2, 5).reshape(b, c, h, w)
        outs.append(o)
    return outs


def lawin_head(sd, feats, training=False, drop_mask=None):
    """modules/lawin.py:181-229.  ``feats`` = 4 NCHW maps (S/4 .. S/32).
    ``drop_mask``: optional [B, E] multiplier standing in for Dropout2d(0.1)
    (:138,228); None = eval / identity.  BN running stats inside ``sd`` are
    updated in place when ``training``."""
    x1, x2, x3, x4 = feats
    h8, w8 = x2.shape[-2:]
    outs = [level_mlp(sd, 'linear_c2', x2)]
    for i, f in enumerate((x3, x4)):
        outs.append(F.interpolate(level_mlp(sd, f'linear_c{i + 3}', f), size=(h8, w8),
                                  mode='bilinear', align_corners=False))
    feat = conv_module(sd, 'linear_fuse', torch.cat(outs[::-1], dim=1), training)
    branches = lawin_branches(sd, feat, training)
    pool = conv_module(sd, 'image_pool.1', F.adaptive_avg_pool2d(feat, 1), training)
    pool = F.interpolate(pool, size=(h8, w8), mode='bilinear', align_corners=False)
    h4, w4 = x1.shape[-2:]
    parts = [F.interpolate(t, size=(h4, w4), mode='bilinear', align_corners=False)
             for t in (feat, *branches, pool)]
    out = conv_module(sd, 'cat', torch.cat(parts, dim=1), training)
    c1 = level_mlp(sd, 'linear_c1', x1)
    fused = conv_module(sd, 'low_level_fuse', torch.cat([out, c1], dim=1), training)
    if drop_mask is not None:
        fused = fused * drop_mask[:, :, None, None]
    return F.conv2d(fused, sd['linear_pred.weight'], sd['linear_pred.bias'])


def segmentor(sd_enc, sd_head, x, variant='B0', training=False, keeps=None, drop_mask=None):
    """The MiT+Lawin composition SURVEY.md §0.2 prescribes: encoder -> head ->
    bilinear (align_corners=False) up-sampling of the S/4 logits to the input size."""
    from .mit import mit_forward
    logits4 = lawin_head(sd_head, mit_forward(sd_enc, x, variant, keeps), training, drop_mask)
    return F.interpolate(logits4, size=x.shape[-2:], mode='bilinear', align_corners=False)
