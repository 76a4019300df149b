"""CPU oracle: ConvNeXt encoder, Decoder and DNet.  TEST INFRASTRUCTURE ONLY.

Plain-PyTorch functional restatement of /root/reference/modules/convnext.py (LayerNorm :6-21, Block :24-51, Stem :54-59,
DownSample :62-67, ConvNeXt :70-130), core/decoder.py:11-59 (with modules/blocks.py:201-242 DeconvolutionBlock and the 1x1
ConvolutionBlock of :17-198), core/dnet.py:9-52 and modules/layers.py:25-43.  Pinned against the reference by
tests/golden/make_golden.py (vectors convnext_x_64.npz, dnet_x_128.npz).
"""
import torch
import torch.nn.functional as F

from .lawin import lawin_head

VARIANTS = {'X': ([2, 2, 4, 2], [32, 64, 128, 256]), 'T': ([3, 3, 9, 3], [96, 192, 384, 768]),
            'S': ([3, 3, 27, 3], [96, 192, 384, 768]), 'B': ([3, 3, 27, 3], [128, 256, 512, 1024])}


def cf_layernorm(sd, p, x, eps=1e-6):
    """modules/convnext.py:16-20 (channel first, biased variance)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return sd[p + '.weight'][:, None, None] * x + sd[p + '.bias'][:, None, None]


def block(sd, p, x, keep=None):
    """modules/convnext.py:37-51."""
    c = x.shape[1]
    y = F.conv2d(x, sd[p + '.dw_conv.weight'], sd[p + '.dw_conv.bias'], padding=3, groups=c).permute(0, 2, 3, 1)
    y = F.layer_norm(y, (c,), sd[p + '.norm.weight'], sd[p + '.norm.bias'], 1e-6)
    y = F.linear(F.gelu(F.linear(y, sd[p + '.pw_conv1.weight'], sd[p + '.pw_conv1.bias'])), sd[p + '.pw_conv2.weight'], sd[p + '.pw_conv2.bias'])
    if p + '.gamma' in sd:
        y = sd[p + '.gamma'] * y
    y = y.permute(0, 3, 1, 2)
    if keep is not None:
        y = y * keep[:, None, None, None]
    return x + y


def convnext_forward(sd, x, variant='T'):
    """modules/convnext.py:122-130."""
    depths, _ = VARIANTS[variant]
    outs = []
    for i in range(4):
        if i == 0:
            x = cf_layernorm(sd, 'downsample_layers.0.1', F.conv2d(x, sd['downsample_layers.0.0.weight'], sd['downsample_layers.0.0.bias'], stride=4))
        else:
            x = cf_layernorm(sd, f'downsample_layers.{i}.0', x)
            x = F.conv2d(x, sd[f'downsample_layers.{i}.1.weight'], sd[f'downsample_layers.{i}.1.bias'], stride=2)
        for j in range(depths[i]):
            x = block(sd, f'stages.{i}.{j}', x)
        outs.append(cf_layernorm(sd, f'norm{i}', x))
    return outs


def decoder_forward(sd, feats, training=False, drop_mask=None):
    """core/decoder.py:49-59; sd keys relative to the Decoder (decoder.*, deconvolution.net.*, feature_reducer.net.p_conv.conv.*)."""
    head = {k[len('decoder.'):]: v for k, v in sd.items() if k.startswith('decoder.')}
    x = lawin_head(head, feats, training, drop_mask)
    a = F.conv_transpose2d(x, sd['deconvolution.net.0.weight'], sd['deconvolution.net.0.bias'], stride=4)
    p = 'deconvolution.net.1'
    if training:
        a = F.batch_norm(a, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'], True, 0.1, 1e-5)
        sd[p + '.num_batches_tracked'] += 1
    else:
        a = F.batch_norm(a, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'], False, 0.1, 1e-5)
    a = F.relu(a)
    b = F.interpolate(x, size=a.shape[-2:], mode='nearest')
    b = F.conv2d(b, sd['feature_reducer.net.p_conv.conv.weight'], sd['feature_reducer.net.p_conv.conv.bias'])
    return a * b


def dnet_forward(sd, x, variant='T', training=False, drop_mask=None):
    """core/dnet.py:36-52: probabilities [B, nc, H, W]."""
    h, w = x.shape[-2:]
    enc = {k[len('encoder.encoder.'):]: v for k, v in sd.items() if k.startswith('encoder.encoder.')}
    dec = {k[len('decoder.'):]: v for k, v in sd.items() if k.startswith('decoder.')}
    y = decoder_forward(dec, convnext_forward(enc, x, variant), training, drop_mask)
    y = F.interpolate(y, size=(h, w), mode='nearest')
    y = F.conv2d(y, sd['terminator.terminator.0.weight'], sd['terminator.terminator.0.bias'])
    return torch.softmax(y, dim=1)
