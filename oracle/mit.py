"""CPU oracle: MiT (SegFormer) encoder.  TEST INFRASTRUCTURE ONLY.

Functional restatement of /root/reference/modules/segformer.py working on a
flat ``state_dict`` (same keys as the reference ``SegFormer.state_dict()``).
"""
import torch
import torch.nn.functional as F

# modules/segformer.py:103-110  (embed dims, depths)
VARIANTS = {
    'B0': ((32, 64, 160, 256), (2, 2, 2, 2)),
    'B1': ((64, 128, 320, 512), (2, 2, 2, 2)),
    'B2': ((64, 128, 320, 512), (3, 4, 6, 3)),
    'B3': ((64, 128, 320, 512), (3, 4, 18, 3)),
    'B4': ((64, 128, 320, 512), (3, 8, 27, 3)),
    'B5': ((64, 128, 320, 512), (3, 6, 40, 3)),
}
HEADS = (1, 2, 5, 8)       # modules/segformer.py:141,150,159,168
SR = (8, 4, 2, 1)
PATCH = ((7, 4), (3, 2), (3, 2), (3, 2))   # modules/segformer.py:128-131


def drop_path_rates(variant):
    """modules/segformer.py:124,133-136."""
    depths = VARIANTS[variant][1]
    return [v.item() for v in torch.linspace(0, 0.1, sum(depths))]


def _ln(sd, p, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[p + '.weight'], sd[p + '.bias'], eps)


def attention(sd, p, x, h, w, head, sr):
    """modules/segformer.py:21-41."""
    b, n, c = x.shape
    d = c // head
    q = F.linear(x, sd[p + '.q.weight'], sd[p + '.q.bias'])
    q = q.view(b, n, head, d).transpose(1, 2)                      # b,head,n,d
    if sr > 1:
        xi = x.transpose(1, 2).reshape(b, c, h, w)
        xi = F.conv2d(xi, sd[p + '.sr.weight'], sd[p + '.sr.bias'], stride=sr)
        x = _ln(sd, p + '.norm', xi.flatten(2).transpose(1, 2))
    kv = F.linear(x, sd[p + '.kv.weight'], sd[p + '.kv.bias'])
    kv = kv.view(b, -1, 2, head, d).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    a = torch.softmax((q @ k.transpose(-2, -1)) * (d ** -0.5), dim=-1)
    o = (a @ v).transpose(1, 2).reshape(b, n, c)
    return F.linear(o, sd[p + '.proj.weight'], sd[p + '.proj.bias'])


def mix_ffn(sd, p, x, h, w):
    """modules/segformer.py:44-64."""
    b, n, _ = x.shape
    y = F.linear(x, sd[p + '.fc1.weight'], sd[p + '.fc1.bias'])
    c = y.shape[-1]
    yi = y.transpose(1, 2).reshape(b, c, h, w)
    yi = F.conv2d(yi, sd[p + '.dw_conv.dw_conv.weight'], sd[p + '.dw_conv.dw_conv.bias'],
                  padding=1, groups=c)
    y = F.gelu(yi.flatten(2).transpose(1, 2))
    return F.linear(y, sd[p + '.fc2.weight'], sd[p + '.fc2.bias'])


def block(sd, p, x, h, w, head, sr, keep=None):
    """modules/segformer.py:96-99.  ``keep``: optional pair of [B] per-sample
    DropPath multipliers (floor(kp+rand)/kp, modules/helpers.py:17-24)."""
    a = attention(sd, p + '.attn', _ln(sd, p + '.norm1', x), h, w, head, sr)
    if keep is not None:
        a = a * keep[0].view(-1, 1, 1)
    x = x + a
    m = mix_ffn(sd, p + '.mlp', _ln(sd, p + '.norm2', x), h, w)
    if keep is not None:
        m = m * keep[1].view(-1, 1, 1)
    return x + m


def patch_embed(sd, p, x, k, s):
    """modules/segformer.py:79-84."""
    x = F.conv2d(x, sd[p + '.proj.weight'], sd[p + '.proj.bias'], stride=s, padding=k // 2)
    h, w = x.shape[-2:]
    return _ln(sd, p + '.norm', x.flatten(2).transpose(1, 2)), h, w


def mit_forward(sd, x, variant='B0', keeps=None):
    """modules/segformer.py:174-200.  Returns 4 NCHW feature maps.
    ``keeps``: optional dict {(stage, blk): (keep_attn[B], keep_mlp[B])}."""
    dims, depths = VARIANTS[variant]
    b = x.shape[0]
    outs = []
    for s in range(4):
        k, st = PATCH[s]
        x, h, w = patch_embed(sd, f'patch_embed{s + 1}', x, k, st)
        for i in range(depths[s]):
            kp = None if keeps is None else keeps.get((s, i))
            x = block(sd, f'block{s + 1}.{i}', x, h, w, HEADS[s], SR[s], kp)
        x = _ln(sd, f'norm{s + 1}', x).reshape(b, h, w, -1).permute(0, 3, 1, 2)
        outs.append(x)
    return tuple(outs)
