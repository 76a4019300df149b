"""state_dict key/shape manifests of the reference modules (checkpoint contract).

Reference: modules/segformer.py:7-172 (SegFormer), modules/lawin.py:7-138 (LawinHead).
The nn.Modules in this package register parameters under exactly these names so
reference checkpoints load unchanged (checkpoint row).
"""

MIT_VARIANTS = {           # modules/segformer.py:103-110
    'B0': ((32, 64, 160, 256), (2, 2, 2, 2)),
    'B1': ((64, 128, 320, 512), (2, 2, 2, 2)),
    'B2': ((64, 128, 320, 512), (3, 4, 6, 3)),
    'B3': ((64, 128, 320, 512), (3, 4, 18, 3)),
    'B4': ((64, 128, 320, 512), (3, 8, 27, 3)),
    'B5': ((64, 128, 320, 512), (3, 6, 40, 3)),
}
MIT_HEADS = (1, 2, 5, 8)
MIT_SR = (8, 4, 2, 1)
MIT_PATCH = ((7, 4), (3, 2), (3, 2), (3, 2))
LAWIN_BRANCHES = ((8, 64), (4, 16), (2, 4))     # (r, heads) modules/lawin.py:122-124


def _ln(out, p, c):
    out[p + '.weight'] = (c,)
    out[p + '.bias'] = (c,)


def _lin(out, p, cin, cout):
    out[p + '.weight'] = (cout, cin)
    out[p + '.bias'] = (cout,)


def mit_shapes(variant='B0', in_channels=3):
    dims, depths = MIT_VARIANTS[variant]
    out = {}
    cin = in_channels
    for s in range(4):
        c, (k, _) = dims[s], MIT_PATCH[s]
        out[f'patch_embed{s + 1}.proj.weight'] = (c, cin, k, k)
        out[f'patch_embed{s + 1}.proj.bias'] = (c,)
        _ln(out, f'patch_embed{s + 1}.norm', c)
        for i in range(depths[s]):
            p = f'block{s + 1}.{i}'
            _ln(out, p + '.norm1', c)
            _lin(out, p + '.attn.q', c, c)
            _lin(out, p + '.attn.kv', c, 2 * c)
            _lin(out, p + '.attn.proj', c, c)
            if MIT_SR[s] > 1:
                out[p + '.attn.sr.weight'] = (c, c, MIT_SR[s], MIT_SR[s])
                out[p + '.attn.sr.bias'] = (c,)
                _ln(out, p + '.attn.norm', c)
            _ln(out, p + '.norm2', c)
            _lin(out, p + '.mlp.fc1', c, 4 * c)
            out[p + '.mlp.dw_conv.dw_conv.weight'] = (4 * c, 1, 3, 3)
            out[p + '.mlp.dw_conv.dw_conv.bias'] = (4 * c,)
            _lin(out, p + '.mlp.fc2', 4 * c, c)
        _ln(out, f'norm{s + 1}', c)
        cin = c
    return out


def _bn(out, p, c):
    out[p + '.weight'] = (c,)
    out[p + '.bias'] = (c,)
    out[p + '.running_mean'] = (c,)
    out[p + '.running_var'] = (c,)
    out[p + '.num_batches_tracked'] = ()


def _conv_module(out, p, cin, cout):
    out[p + '.conv.weight'] = (cout, cin, 1, 1)
    _bn(out, p + '.bn', cout)


def lawin_shapes(in_channels, embed_dim=512, out_channels=8):
    e = embed_dim
    out = {}
    for i, c in enumerate(in_channels):
        _lin(out, f'linear_c{i + 1}.proj', c, 48 if i == 0 else e)
    _conv_module(out, 'linear_fuse', 3 * e, e)
    for r, heads in LAWIN_BRANCHES:
        p = f'lawin_{r}'
        for h in range(heads):
            _lin(out, f'{p}.position_mixing.{h}', 64, 64)
        inter = max(e // 2, 1)
        for nm in ('g', 'theta', 'phi'):
            out[f'{p}.{nm}.weight'] = (inter, e, 1, 1)
            out[f'{p}.{nm}.bias'] = (inter,)
        out[f'{p}.conv_out.0.weight'] = (e, inter, 1, 1)
        _bn(out, f'{p}.conv_out.1', e)
    _conv_module(out, 'image_pool.1', e, e)
    _conv_module(out, 'cat', 5 * e, e)
    _conv_module(out, 'low_level_fuse', e + 48, e)
    out['linear_pred.weight'] = (out_channels, e, 1, 1)
    out['linear_pred.bias'] = (out_channels,)
    return out


CONVNEXT_VARIANTS = {      # modules/convnext.py:72-77
    'X': ((2, 2, 4, 2), (32, 64, 128, 256)), 'T': ((3, 3, 9, 3), (96, 192, 384, 768)),
    'S': ((3, 3, 27, 3), (96, 192, 384, 768)), 'B': ((3, 3, 27, 3), (128, 256, 512, 1024)),
}


def convnext_shapes(variant='T', in_channels=3):
    """modules/convnext.py:79-130."""
    depths, dims = CONVNEXT_VARIANTS[variant]
    out = {'downsample_layers.0.0.weight': (dims[0], in_channels, 4, 4), 'downsample_layers.0.0.bias': (dims[0],)}
    _ln(out, 'downsample_layers.0.1', dims[0])
    for i in range(1, 4):
        _ln(out, f'downsample_layers.{i}.0', dims[i - 1])
        out[f'downsample_layers.{i}.1.weight'] = (dims[i], dims[i - 1], 2, 2)
        out[f'downsample_layers.{i}.1.bias'] = (dims[i],)
    for i in range(4):
        c = dims[i]
        for j in range(depths[i]):
            p = f'stages.{i}.{j}'
            out[p + '.gamma'] = (c,)
            out[p + '.dw_conv.weight'] = (c, 1, 7, 7)
            out[p + '.dw_conv.bias'] = (c,)
            _ln(out, p + '.norm', c)
            _lin(out, p + '.pw_conv1', c, 4 * c)
            _lin(out, p + '.pw_conv2', 4 * c, c)
        _ln(out, f'norm{i}', c)
    return out


def decoder_shapes(in_channels, out_channels, embed_dim=512):
    """core/decoder.py:11-47."""
    out = {'decoder.' + k: v for k, v in lawin_shapes(in_channels, embed_dim, 2 * out_channels).items()}
    out['deconvolution.net.0.weight'] = (2 * out_channels, out_channels, 3, 3)
    out['deconvolution.net.0.bias'] = (out_channels,)
    _bn(out, 'deconvolution.net.1', out_channels)
    out['feature_reducer.net.p_conv.conv.weight'] = (out_channels, 2 * out_channels, 1, 1)
    out['feature_reducer.net.p_conv.conv.bias'] = (out_channels,)
    return out


def dnet_shapes(image_channels=3, num_classes=6, embed_dim=256, stem_code='T'):
    """core/dnet.py:9-34."""
    out = {'encoder.encoder.' + k: v for k, v in convnext_shapes(stem_code, image_channels).items()}
    dims = list(CONVNEXT_VARIANTS[stem_code][1])
    out.update({'decoder.' + k: v for k, v in decoder_shapes(dims, 2 * num_classes, embed_dim).items()})
    out['terminator.terminator.0.weight'] = (num_classes, 2 * num_classes, 1, 1)
    out['terminator.terminator.0.bias'] = (num_classes,)
    return out
