"""MiT + Lawin segmentor: the 5-line composition SURVEY.md §0.2 says the reference never wires up itself
(modules/segformer.py:102 SegFormer is imported by nothing; modules/lawin.py:224-229 LawinHead returns S/4 logits),
with bilinear (align_corners=False) up-sampling of the logits to the input size and the ``out_channels`` attribute
``core/igniter.py:43`` requires, so ``LightningSemSeg(model=SegFormerLawin(...))`` runs unchanged."""
import torch
from torch import nn, Tensor

from . import ops
from .runtime import as_tokens_nhwc
from .segformer import SegFormer
from .lawin import LawinHead


class UpsampleLogitsFn(torch.autograd.Function):
    """F.interpolate(x, size, mode='bilinear', align_corners=False) on channels-last logits."""

    @staticmethod
    def forward(ctx, x, size):
        b, c, h, w = x.shape
        t = as_tokens_nhwc(x.detach(), x.dtype)
        y = ops.bilinear_fwd(t, b, h, w, size[0], size[1])
        ctx.geom = (b, c, h, w, size[0], size[1])
        return y.view(b, size[0], size[1], c).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        b, c, h, w, ho, wo = ctx.geom
        d = as_tokens_nhwc(dy, dy.dtype)
        dx = ops.bilinear_bwd(d.contiguous(), b, h, w, ho, wo)
        return dx.view(b, h, w, c).permute(0, 3, 1, 2), None


class SegFormerLawin(nn.Module):
    def __init__(self, model_name: str = 'B2', embed_dim: int = 512, num_classes: int = 6, in_channels: int = 3):
        super().__init__()
        self.encoder = SegFormer(model_name, in_channels)
        self.decoder = LawinHead(self.encoder.channels, embed_dim, num_classes)
        self.out_channels = num_classes

    def forward(self, x: Tensor) -> Tensor:
        logits4 = self.decoder(self.encoder(x))
        return UpsampleLogitsFn.apply(logits4, tuple(x.shape[-2:]))
