"""data_factory/utils.py:6-37 on the device: the step right before the forward.

The reference converts every tile to float32 on the CPU workers (``image_to_tensor``: ``(x - mu) / sigma``, with ``sigma = 255`` for
byte images) and ships 4 bytes per value to the GPU; the labels travel as int64.  Here the batch is uploaded as uint8 and one
kernel per tensor does the arithmetic on the device: ``image_to_tensor(..., dtype=torch.float32)`` returns the reference tensor bit
for bit (NCHW), and with the engine's compute dtype it returns the permuted view of a channels-last map, which the first layer
of every model here reads without another conversion pass.  No CPU fallback: the inputs must live on a HIP device.
"""
import numpy as np
import torch

from . import ops
from .runtime import compute_dtype_for


_stats = {}


def _per_channel(mu, sigma, c, device):
    """mu / sigma as float32 [C] device tensors (uploaded once per distinct pair), checked like data_factory/utils.py:20,29-31."""
    mu = [0] * c if mu is None else list(mu)
    sigma = [255] * c if sigma is None else list(sigma)
    assert len(mu) == len(sigma) == c, "Invalid 'mu', 'sigma'"
    key = (tuple(float(v) for v in mu), tuple(float(v) for v in sigma), device)
    t = _stats.get(key)
    if t is None:
        m = torch.tensor(np.asarray(mu, dtype=np.float64), dtype=torch.float32)          # torch.tensor(np.array(mu), dtype=float32), as :21-28
        s = torch.tensor(np.asarray(sigma, dtype=np.float64), dtype=torch.float32)
        assert bool(torch.all(s > 0)), "'sigma' cannot have zero or negative values"
        if len(_stats) > 64:
            _stats.clear()
        t = _stats[key] = (m.to(device), s.to(device))
    return t


def image_to_tensor(arr, mu=None, sigma=None, dtype=None):
    """arr: uint8 device tensor [C, H, W] (the reference signature) or [B, C, H, W]; mu / sigma: per-channel sequences, defaults 0
    and 255 as data_factory/utils.py:13-19.  dtype None = the engine's compute dtype (channels-last storage); torch.float32 = the
    reference's dense NCHW float tensor."""
    single = arr.dim() == 3
    x = arr.unsqueeze(0) if single else arr
    if not x.is_cuda:
        raise RuntimeError('semseg_amd: tensors must live on a HIP device (no CPU fallback)')
    m, s = _per_channel(mu, sigma, x.shape[1], x.device)
    if dtype is None:
        dtype = compute_dtype_for(torch.empty(0, dtype=torch.float32, device=x.device))
    out = ops.stage_image(x.contiguous(), m, s, dtype, channels_last=dtype != torch.float32)
    return out[0] if single else out


def label_to_tensor(arr):
    """uint8 device tensor -> int64 (data_factory/utils.py:36-37)."""
    return ops.stage_label(arr.contiguous())
