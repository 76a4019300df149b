"""Deterministic, torch-version-independent synthetic weights.

Every tensor is drawn from a numpy PCG64 stream keyed by the crc32 of its
state_dict key, so the same ``state_dict`` layout always receives the same
values (used by the golden-vector generator, the parity tests and bench.py;
there is no network for real checkpoints).
"""
import zlib
import numpy as np
import torch


def det_tensor(name, shape, kind=None):
    rng = np.random.Generator(np.random.PCG64(zlib.crc32(name.encode())))
    shape = tuple(shape)
    leaf = name.rsplit('.', 1)[-1]
    if leaf == 'num_batches_tracked':
        return np.zeros(shape, dtype=np.int64)
    if leaf == 'running_var':
        return (1.0 + 0.2 * rng.random(shape)).astype(np.float32)
    if leaf == 'running_mean':
        return (0.1 * rng.standard_normal(shape)).astype(np.float32)
    if leaf == 'gamma':                       # ConvNeXt layer scale
        return (0.5 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
    if len(shape) <= 1:
        is_scale = leaf == 'weight'           # norm scale vectors
        base = 1.0 if is_scale else 0.0
        return (base + 0.1 * rng.standard_normal(shape)).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    return (rng.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)


def det_state_dict(shapes):
    """shapes: {key: shape}  ->  {key: torch tensor}."""
    return {k: torch.from_numpy(np.asarray(det_tensor(k, s))) for k, s in sorted(shapes.items())}


def det_fill_(module, prefix=''):
    """Fill a module's parameters and buffers in place; returns the module."""
    sd = module.state_dict()
    new = det_state_dict({k: tuple(v.shape) for k, v in sd.items()})
    with torch.no_grad():
        for k, v in sd.items():
            v.copy_(new[k].to(v.dtype).reshape(v.shape))
    return module


def det_input(name, shape, lo=0.0, hi=1.0):
    rng = np.random.Generator(np.random.PCG64(zlib.crc32(name.encode())))
    return torch.from_numpy((lo + (hi - lo) * rng.random(tuple(shape))).astype(np.float32))


def det_labels(name, shape, num_classes, cell=4):
    """Piecewise-constant label map (UAVPal-shaped): random class per cell x cell block."""
    rng = np.random.Generator(np.random.PCG64(zlib.crc32(name.encode())))
    b, h, w = shape
    small = rng.integers(0, num_classes, size=(b, -(-h // cell), -(-w // cell)))
    return torch.from_numpy(np.repeat(np.repeat(small, cell, 1), cell, 2)[:, :h, :w].astype(np.int64))
