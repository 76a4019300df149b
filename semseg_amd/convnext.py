"""ConvNeXt encoder on the MI355X kernels.

Same constructors / forward signatures / ``.channels`` / ``state_dict`` layout as the reference ``modules/convnext.py``
(LayerNorm :6-21, Block :24-51, Stem :54-59, DownSample :62-67, ConvNeXt :70-130).  Activations are channels-last token
matrices, so the reference's NCHW<->NHWC permutes (:40,49) disappear and its channel-first LayerNorm (:16-20: biased variance
over C) is the same kernel as nn.LayerNorm on tokens.  Per block: depthwise 7x7 (``ss_dwconvk_fwd``) -> LayerNorm(eps 1e-6)
-> pw_conv1 GEMM -> GELU -> pw_conv2 GEMM with the layer scale folded into its weight (gamma * (W x + b) == (gamma W) x +
gamma b) and the residual (+ DropPath row scale) fused into its epilogue.  The 4x4/s4 stem and the 2x2/s2 down-sampling
convolutions are non-overlapping patch GEMMs (im2col is a pure re-layout).
"""
import torch
from torch import nn, Tensor

from . import ops
from .runtime import compute_dtype_for, as_tokens_nhwc, drop_path_keep
from .segformer import _ParamFn, w_fwd, w_bwd, w_taps, linear_grads
from . import weightprep
from .helpers import DropPath

CONVNEXT_VARIANTS = {                       # modules/convnext.py:72-77  [depths, dims, drop-path rate]
    'X': [[2, 2, 4, 2], [32, 64, 128, 256], 0.0],
    'T': [[3, 3, 9, 3], [96, 192, 384, 768], 0.0],
    'S': [[3, 3, 27, 3], [96, 192, 384, 768], 0.0],
    'B': [[3, 3, 27, 3], [128, 256, 512, 1024], 0.0],
}


def _nchw_view(tok, b, h, w):
    return tok.view(b, h, w, -1).permute(0, 3, 1, 2)


class _LNFn(_ParamFn):
    """channel-first LayerNorm (modules/convnext.py:6-21) == LayerNorm over C of the token matrix."""

    @staticmethod
    def fwd(P, x, meta):
        y, m, r = ops.layernorm_fwd(x, P['weight'], P['bias'], meta)
        return y, {'x': x, 'm': m, 'r': r}

    @staticmethod
    def bwd(P, S, dy, meta):
        dx, dgb = ops.layernorm_bwd(dy, S['x'], S['m'], S['r'], P['weight'])
        return dx, {'weight': dgb[0], 'bias': dgb[1]}


class _PatchConvFn(_ParamFn):
    """nn.Conv2d(c1, c2, k, s) with k == s and no padding (stem 4x4/s4, down-sampling 2x2/s2): a GEMM over the image's own patches
    where the tile kernels can address them (ss_conv_patch_*: the down-sampling layers), else im2col + GEMM (the 3-channel stem)."""

    @staticmethod
    def fwd(P, x, meta):
        B, H, W, Cin, k = meta
        wp = weightprep.pack(P['weight'], x.dtype)
        if ops.conv_patch_ok(x, B, H, W, Cin, k, wp.shape[0]):
            return ops.conv_patch_fwd(x, B, H, W, Cin, k, wp, bias=P['bias']), {'x': x}
        col, ho, wo = ops.im2col(x, B, H, W, Cin, k, k, 0)
        return ops.gemm_nt(col, wp, bias=P['bias']), {'col': col}

    @staticmethod
    def bwd(P, S, dy, meta):
        B, H, W, Cin, k = meta
        wpt = weightprep.pack(P['weight'], dy.dtype, transposed=True) if Cin % 8 == 0 else None
        if 'x' in S:
            gp, db = ops.conv_patch_wgrad(dy, S['x'], B, H, W, Cin, k)
            dx = ops.conv_patch_dgrad(dy, wpt, B, H, W, Cin, k)
        else:
            gp, db = ops.gemm_tn(dy, S['col'], colsum=True)
            dx = None                                            # the RGB image needs no gradient
            if wpt is not None:
                dx = ops.col2im(ops.gemm_nt(dy, wpt), B, H, W, Cin, k, k, 0)
        return dx, {'bias': db, 'weight': ops.unpack_conv_wgrad(gp, dy.shape[1], Cin, k)}


class _BlockFn(_ParamFn):
    @staticmethod
    def fwd(P, x, meta):
        B, h, w, keep = meta
        T = x.dtype
        C = x.shape[1]
        w49 = w_taps(P['dw_conv.weight'])
        d = ops.dwconvk_fwd(x, w49, P['dw_conv.bias'], B, h, w, 7)
        n, m, r = ops.layernorm_fwd(d, P['norm.weight'], P['norm.bias'], 1e-6)
        z1, a = ops.gemm_nt_gelu2(n, w_fwd(P['pw_conv1.weight'], T), bias=P['pw_conv1.bias'])     # pre-activation and GELU from one epilogue
        gamma = P.get('gamma')
        w2, b2 = P['pw_conv2.weight'].detach(), P['pw_conv2.bias'].detach()
        if gamma is not None:                                # fold the layer scale into the projection (parameter-sized work)
            w2 = ops.rowscale(w2.contiguous(), gamma.detach().contiguous(), 1)
            b2 = (gamma.detach() * b2).contiguous()
        out = ops.gemm_nt(a, w_fwd(w2, T), bias=b2, res=x, rowscale=keep, rows_per_scale=h * w if keep is not None else 0)
        return out, {'x': x, 'd': d, 'm': m, 'r': r, 'n': n, 'z1': z1, 'a': a, 'w2s': w2}

    @staticmethod
    def bwd(P, S, dout, meta):
        B, h, w, keep = meta
        T = dout.dtype
        C = dout.shape[1]
        g = dout if keep is None else ops.rowscale(dout, keep, h * w)
        gw, cs = linear_grads(g, S['a'])                      # gradients of the FOLDED projection
        gamma = P.get('gamma')
        G = {}
        if gamma is not None:
            gd = gamma.detach().contiguous()
            G['pw_conv2.weight'] = ops.rowscale(gw, gd, 1)
            G['pw_conv2.bias'] = gd * cs
            G['gamma'] = ops.rowdot_f32(gw, P['pw_conv2.weight'].detach().contiguous()) + cs * P['pw_conv2.bias'].detach()
        else:
            G['pw_conv2.weight'], G['pw_conv2.bias'] = gw, cs
        dz1 = ops.gemm_nt(g, w_bwd(S['w2s'], T), res=S['z1'], act=ops.GELU_GRAD)    # (g W2) * GELU'(z1) in the GEMM epilogue
        G['pw_conv1.weight'], G['pw_conv1.bias'] = linear_grads(dz1, S['n'])
        dn = ops.gemm_nt(dz1, w_bwd(P['pw_conv1.weight'], T))
        dd, dgb = ops.layernorm_bwd(dn, S['d'], S['m'], S['r'], P['norm.weight'])
        G['norm.weight'], G['norm.bias'] = dgb[0], dgb[1]
        dwdb = ops.dwconvk_wgrad(dd, S['x'], B, h, w, 7)
        G['dw_conv.weight'] = ops.transpose_cast(dwdb[:49].contiguous(), torch.float32).reshape(C, 1, 7, 7)
        G['dw_conv.bias'] = dwdb[49]
        flip = w_taps(P['dw_conv.weight'], reverse=True)
        dx = ops.dwconvk_fwd(dd, flip, None, B, h, w, 7, res=dout if dout.is_contiguous() else dout.contiguous())   # + the skip path's gradient
        return dx, G


# ------------------------------------------------------------------ nn.Modules with the reference's API
class LayerNorm(nn.Module):
    """Channel first layer norm"""

    def __init__(self, normalized_shape, eps=1e-6) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps

    def forward(self, x: Tensor) -> Tensor:
        b, c, h, w = x.shape
        y = _LNFn.run(as_tokens_nhwc(x, compute_dtype_for(x)), self.eps, self)
        return _nchw_view(y, b, h, w)


class Block(nn.Module):
    def __init__(self, dim, dpr=0., init_value=1e-6):
        super().__init__()
        self.dw_conv = nn.Conv2d(dim, dim, (7, 7), (1, 1), 3, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.pw_conv1 = nn.Linear(dim, 4 * dim)
        self.act = nn.GELU()
        self.pw_conv2 = nn.Linear(4 * dim, dim)
        self.gamma = nn.Parameter(init_value * torch.ones((dim,)), requires_grad=True) if init_value > 0 else None
        self.drop_path = DropPath(dpr) if dpr > 0. else nn.Identity()
        self.dpr = dpr

    def forward(self, x: Tensor) -> Tensor:
        b, c, h, w = x.shape
        T = compute_dtype_for(x)
        keep = drop_path_keep(b, self.dpr, T, x.device) if (self.training and self.dpr > 0.) else None
        y = _BlockFn.run(as_tokens_nhwc(x, T), (b, h, w, keep), self)
        return _nchw_view(y, b, h, w)


class _Conv2dPatch(nn.Conv2d):
    """nn.Conv2d(c1, c2, k, s) with k == s (keeps the reference's parameter names '0.weight' / '1.weight' in Stem / DownSample)."""

    def forward(self, x: Tensor) -> Tensor:
        b, c, h, w = x.shape
        k = self.kernel_size[0]
        if h % k or w % k:
            raise ValueError(f'ConvNeXt patch convolution {k}x{k}/s{k}: input {h}x{w} is not a multiple of {k}')
        y = _PatchConvFn.run(as_tokens_nhwc(x, compute_dtype_for(x)), (b, h, w, c, k), self)
        return _nchw_view(y, b, h // k, w // k)


class Stem(nn.Sequential):
    def __init__(self, c1, c2, k, s):
        assert k == s, 'Stem: the MI355X path implements the non-overlapping (k == s) patch convolution the reference uses'
        super().__init__(_Conv2dPatch(c1, c2, k, s), LayerNorm(c2))


class DownSample(nn.Sequential):
    def __init__(self, c1, c2, k, s):
        assert k == s, 'DownSample: the MI355X path implements the non-overlapping (k == s) patch convolution the reference uses'
        super().__init__(LayerNorm(c1), _Conv2dPatch(c1, c2, k, s))


class ConvNeXt(nn.Module):
    __available_configs = CONVNEXT_VARIANTS

    def __init__(self, model_name: str = 'T', in_channels: int = 3) -> None:
        super().__init__()
        assert model_name in self.__available_configs.keys(), (
            f"Unknown ConvNeXt variant '{model_name}'! Supported variants:{list(self.__available_configs.keys())}")
        assert in_channels >= 1, "Input channels must be a positive integer!"
        depths, embed_dims, drop_path_rate = self.__available_configs[model_name]
        self.channels = embed_dims
        self.downsample_layers = nn.ModuleList([
            Stem(in_channels, embed_dims[0], 4, 4),
            *[DownSample(embed_dims[i], embed_dims[i + 1], 2, 2) for i in range(3)]])
        self.stages = nn.ModuleList()
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        cur = 0
        for i in range(4):
            self.stages.append(nn.Sequential(*[Block(embed_dims[i], dpr[cur + j]) for j in range(depths[i])]))
            cur += depths[i]
        for i in range(4):
            self.add_module(f"norm{i}", LayerNorm(embed_dims[i]))

    def forward(self, x: Tensor):
        outs = list()
        for i in range(4):
            x = self.downsample_layers[i](x)
            x = self.stages[i](x)
            outs.append(getattr(self, f"norm{i}")(x))
        return outs
