"""Small modules shared by the networks (reference: modules/helpers.py)."""
import torch
from torch import nn, Tensor


class DropPath(nn.Module):
    """Stochastic depth per sample (modules/helpers.py:9-24).  Inside ``segformer.Block`` the multiplier is fused into
    the GEMM epilogues; this standalone module keeps the reference behaviour for other callers."""

    def __init__(self, drop_prob: float = None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x: Tensor) -> Tensor:
        if self.drop_prob == 0. or not self.training:
            return x
        kp = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        random_tensor = kp + torch.rand(shape, dtype=x.dtype, device=x.device)
        random_tensor.floor_()
        return x.div(kp) * random_tensor
