"""core/decoder.py Decoder, core/dnet.py DNet, core/image_encoder.py ImageEncoder and modules/layers.py MultilabelClassifier on the
MI355X kernels (same constructors, forward signatures and ``state_dict`` keys).

``Decoder``: LawinHead -> {ConvTranspose2d k3 s4 + BatchNorm2d + ReLU} x {nearest-up + 1x1 conv}; the transposed convolution has
stride > kernel, so it is ONE GEMM (x . W[(o,dy,dx)][c]) whose columns are stamped as disjoint 3x3 blocks, and the 1x1 conv commutes
with the nearest up-sampling (``csrc/decoder.hip``).  Output (4H4-1) x (4W4-1), as the reference.
``DNet``: ConvNeXt -> Decoder -> nearest resize to the input size -> 1x1 conv -> Softmax2d (probabilities).
"""
from typing import Sequence
import torch
from torch import nn, Tensor

from . import ops, weightprep
from .runtime import compute_dtype_for, as_tokens_nhwc
from .segformer import _ParamFn, w_fwd
from .lawin import LawinHead, _bn_stats
from .convnext import ConvNeXt


def _nchw_view(tok, b, h, w):
    return tok.view(b, h, w, -1).permute(0, 3, 1, 2)


class DeconvolutionBlock(nn.Module):
    """Parameter container with the layout of modules/blocks.py:201-242 (net.0 ConvTranspose2d, net.1 BatchNorm2d, net.2 activation).
    Only the configuration core/decoder.py uses is on the MI355X path: ks 3, stride 4, no padding, groups 1, ReLU."""

    def __init__(self, inc: int, outc: int, ks=(3, 3), stride=(1, 1), dilation=(1, 1), padding: int = 0, output_padding: int = 0,
                 groups: int = 1, bias: bool = True, activation: nn.Module = nn.Mish(True)):
        super().__init__()
        ks = (ks,) * 2 if isinstance(ks, int) else tuple(ks)
        stride = (stride,) * 2 if isinstance(stride, int) else tuple(stride)
        dilation = (dilation,) * 2 if isinstance(dilation, int) else tuple(dilation)
        self.net = nn.Sequential(
            nn.ConvTranspose2d(inc, outc, kernel_size=ks, dilation=dilation, stride=stride, padding=padding, output_padding=output_padding,
                               groups=groups, bias=bias),
            nn.BatchNorm2d(outc), activation)
        self.supported = (ks == (3, 3) and stride == (4, 4) and dilation == (1, 1) and padding == 0 and output_padding == 0 and groups == 1
                          and bias and isinstance(activation, nn.ReLU))


class ConvolutionBlock1x1(nn.Module):
    """modules/blocks.py:17-198 for the 1x1 / stride 1 / bias / no norm / no activation case core/decoder.py:38-47 builds
    (parameter names net.p_conv.conv.{weight,bias})."""

    def __init__(self, inc: int, outc: int):
        super().__init__()
        p_conv = nn.Sequential()
        p_conv.add_module('conv', nn.Conv2d(inc, outc, (1, 1), bias=True))
        self.net = nn.Sequential()
        self.net.add_module('p_conv', p_conv)


class _DecoderTailFn(_ParamFn):
    @staticmethod
    def fwd(P, t, meta):
        B, H4, W4, oc, training, bufs = meta
        T = t.dtype
        inc = t.shape[1]
        ld9 = (9 * oc + 7) // 8 * 8
        # [(o,dy,dx)][c], zero rows pad N to a multiple of 8: a remembered, batch-refreshed copy (no per-call torch.zeros / slice copy)
        wd = weightprep.transpose_padded(P['deconvolution.net.0.weight'], inc, 9 * oc, T, ld9)
        wr = w_fwd(P['feature_reducer.net.p_conv.conv.weight'], T)
        z9 = ops.gemm_nt(t, wd)
        bs = ops.gemm_nt(t, wr, bias=P['feature_reducer.net.p_conv.conv.bias'])
        dbias = P['deconvolution.net.0.bias'].detach()
        if training:
            stats = ops.decoder_stats(z9, B, H4, W4, oc, dbias, 1e-5, 0.1, bufs['running_mean'], bufs['running_var'], bufs['num_batches_tracked'])
        else:
            stats = ops.bn_eval_stats(bufs['running_mean'], bufs['running_var']).reshape(-1)
        prod = ops.decoder_combine_fwd(z9, bs, dbias, stats, P['deconvolution.net.1.weight'], P['deconvolution.net.1.bias'], B, H4, W4, oc)
        return prod, {'t': t, 'wd': wd, 'wr': wr, 'z9': z9, 'bs': bs, 'stats': stats}

    @staticmethod
    def bwd(P, S, dprod, meta):
        B, H4, W4, oc, training, bufs = meta
        t, T = S['t'], dprod.dtype
        inc = t.shape[1]
        dz9, dbs, dgb, dbg = ops.decoder_combine_bwd(dprod, S['z9'], S['bs'], P['deconvolution.net.0.bias'].detach(), S['stats'],
                                                     P['deconvolution.net.1.weight'], P['deconvolution.net.1.bias'], training, B, H4, W4, oc)
        gwd = ops.gemm_tn(dz9, t)[:9 * oc].contiguous()                                    # [(o,dy,dx)][c]
        G = {'deconvolution.net.0.weight': ops.transpose_cast(gwd, torch.float32).reshape(inc, oc, 3, 3),
             'deconvolution.net.0.bias': dbg,
             'deconvolution.net.1.weight': dgb[:oc], 'deconvolution.net.1.bias': dgb[oc:],
             'feature_reducer.net.p_conv.conv.weight': ops.gemm_tn(dbs, t),
             'feature_reducer.net.p_conv.conv.bias': ops.colsum_any(dbs)}
        f32 = (lambda w: w) if T == torch.float32 else (lambda w: ops.cast(w, torch.float32))
        dt = ops.gemm_nt(dz9, ops.transpose_cast(f32(S['wd']), T))
        dt = ops.gemm_nt(dbs, ops.transpose_cast(f32(S['wr'].contiguous()), T), res=dt)
        return dt, G


class Decoder(nn.Module):
    def __init__(self, in_channels: list, out_channels: int, embed_dim: int = 512):
        super().__init__()
        self.decoder = LawinHead(in_channels=in_channels, embed_dim=embed_dim, out_channels=(2 * out_channels))
        self.deconvolution = DeconvolutionBlock(inc=(2 * out_channels), outc=out_channels, ks=(3, 3), stride=(4, 4), dilation=(1, 1), padding=0,
                                                output_padding=0, groups=1, bias=True, activation=nn.ReLU(True))
        self.feature_reducer = ConvolutionBlock1x1(inc=(2 * out_channels), outc=out_channels)
        self.out_channels = out_channels

    def forward(self, features: Sequence[Tensor]) -> Tensor:
        x = self.decoder(features)
        b, c, h4, w4 = x.shape
        if c % 8:
            raise ValueError('Decoder: 2*out_channels must be a multiple of 8 on the MI355X path')
        bn = self.deconvolution.net[1]
        bufs = {'running_mean': bn.running_mean, 'running_var': bn.running_var, 'num_batches_tracked': bn.num_batches_tracked}
        names = [n for n, _ in self.named_parameters() if not n.startswith('decoder.')]
        params = [p for n, p in self.named_parameters() if not n.startswith('decoder.')]
        t = as_tokens_nhwc(x, compute_dtype_for(x))
        prod = _DecoderTailFn.apply(t, (b, h4, w4, self.out_channels, self.training, bufs), names, *params)
        return _nchw_view(prod, b, 4 * h4 - 1, 4 * w4 - 1)


class MultilabelClassifier(nn.Module):
    def __init__(self, in_channels: int, n_classes):
        super().__init__()
        self.terminator = nn.Sequential(nn.Conv2d(in_channels, n_classes, (1, 1), bias=True), nn.Softmax2d())

    def forward(self, x: Tensor, size=None) -> Tensor:
        """x [B, C, Hd, Wd] -> probabilities [B, n_classes, H, W]; ``size`` fuses a preceding nearest resize (core/dnet.py:45-49)."""
        b, c, hd, wd = x.shape
        h, w = (hd, wd) if size is None else size
        t = as_tokens_nhwc(x, compute_dtype_for(x))
        return _TerminatorFn.run(t, (b, hd, wd, h, w), self)


class _TerminatorFn(_ParamFn):
    @staticmethod
    def fwd(P, prod, meta):
        B, Hd, Wd, H, W = meta
        wt = P['terminator.0.weight'].detach().reshape(P['terminator.0.weight'].shape[0], -1).contiguous()
        # the probabilities leave in fp32 whatever the compute dtype: torch.autocast runs softmax in fp32, so the reference's
        # reduced-precision path hands fp32 probabilities to its loss (6 channels: the extra traffic is ~0.2 ms per step)
        probs = ops.terminator_fwd(prod.contiguous(), wt, P['terminator.0.bias'].detach(), B, Hd, Wd, H, W, probs_f32=True)
        return probs, {'prod': prod.contiguous(), 'wt': wt}

    @staticmethod
    def bwd(P, S, dprobs, meta):
        B, Hd, Wd, H, W = meta
        dprod, dw, db = ops.terminator_bwd(dprobs.float().contiguous(), S['prod'], S['wt'], P['terminator.0.bias'].detach(), B, Hd, Wd, H, W)
        return dprod, {'terminator.0.weight': dw, 'terminator.0.bias': db}


class ImageEncoder(nn.Module):
    def __init__(self, in_channels: int, variant_code: str = 'T'):
        super().__init__()
        self.encoder = ConvNeXt(model_name=variant_code, in_channels=in_channels)
        self.channels = self.encoder.channels

    def forward(self, x: Tensor):
        return self.encoder(x)


class DNet(nn.Module):
    def __init__(self, image_channels: int, num_classes: int, embed_dim: int = 256, stem_code: str = 'T'):
        super().__init__()
        self.in_channels = image_channels
        self.out_channels = num_classes
        self.encoder = ImageEncoder(in_channels=image_channels, variant_code=stem_code)
        self.decoder = Decoder(in_channels=self.encoder.channels, out_channels=(2 * num_classes), embed_dim=embed_dim)
        self.terminator = MultilabelClassifier(in_channels=(2 * num_classes), n_classes=num_classes)

    def forward(self, x: Tensor) -> Tensor:
        h, w = x.size(-2), x.size(-1)
        x = self.decoder(self.encoder(x))
        return self.terminator(x, size=(h, w))          # nearest resize to (h, w) fused into the classifier kernel
