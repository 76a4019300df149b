"""data_factory/transforms.py on the device: the geometric augmentations of train.py:68-113.

``hflip`` / ``vflip`` (train.py:18-19,69-78), ``RandomAffine`` (transforms.py:168-322), ``RandomPerspective`` (:325-475) and the
``SegmentationTransform`` scheduler (:40-147) with the reference's constructors, call protocol (generators that yield an
(image_transform, label_transform) pair) and random draws (same ``random`` / ``torch.randint`` calls in the same order, so a seeded
run picks the same parameters).  The transforms act on HIP tensors -- uint8 tiles before ``datastage.image_to_tensor``, float
images, int64 labels, [..., H, W] -- through ``ss_flip`` / ``ss_warp_nearest``; there is no CPU path.

The resampling arithmetic is torchvision's (``torchvision.transforms.functional.affine`` / ``perspective`` on tensors), which the
reference imports but does not vendor: its published algorithm is restated (inverse affine matrix, perspective coefficients by
least squares, float32 sampling grid, ``grid_sample`` nearest / zeros / align_corners=False, fill where the source is outside).
Parity unpinned (no torchvision here, no fixture in the reference); pinned against ``oracle/augment.py``.  Only
``InterpolationMode.NEAREST`` (what train.py uses everywhere) is built.
"""
import math
import random
from functools import partial
from random import choice as choose
from random import random as toss
from typing import Callable, Sequence, Union

import numpy as np
import torch

from . import ops

NEAREST = 'nearest'


def _is_nearest(interpolation):
    return interpolation is None or str(getattr(interpolation, 'value', interpolation)).lower() == NEAREST


def hflip(img: torch.Tensor) -> torch.Tensor:
    return ops.flip(img.contiguous(), True)


def vflip(img: torch.Tensor) -> torch.Tensor:
    return ops.flip(img.contiguous(), False)


# ------------------------------------------------------------------ host-side parameter math (float64, as torchvision's Python)
def inverse_affine_matrix(center, angle, translate, scale, shear):
    """torchvision _get_inverse_affine_matrix: output pixel -> input pixel, both relative to the image centre."""
    rot, sx, sy = math.radians(angle), math.radians(shear[0]), math.radians(shear[1])
    cx, cy = center
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d / scale, -b / scale, 0.0, -c / scale, a / scale, 0.0]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return m


def perspective_coeffs(startpoints, endpoints):
    """torchvision _get_perspective_coeffs: the 8 coefficients mapping an OUTPUT pixel to its source, least squares in float64."""
    a = np.zeros((8, 8), dtype=np.float64)
    for i, (p1, p2) in enumerate(zip(endpoints, startpoints)):
        a[2 * i] = [p1[0], p1[1], 1, 0, 0, 0, -p2[0] * p1[0], -p2[0] * p1[1]]
        a[2 * i + 1] = [0, 0, 0, p1[0], p1[1], 1, -p2[1] * p1[0], -p2[1] * p1[1]]
    b = np.asarray(startpoints, dtype=np.float64).reshape(8)
    return [float(v) for v in np.linalg.lstsq(a, b, rcond=None)[0]]


def _fill_bits(fill, dtype, channels):
    """per-channel fill values as bit patterns of `dtype` (integer dtypes: rounded, as torchvision's cast back)."""
    if fill is None:
        fill = 0
    vals = [fill] * max(1, channels) if isinstance(fill, (int, float)) else list(fill)
    if len(vals) not in (1, channels):
        raise ValueError(f"The number of elements in 'fill' ({len(vals)}) does not match the number of channels ({channels})")
    if len(vals) > 16:
        raise ValueError('at most 16 fill values')
    t = torch.tensor(vals, dtype=torch.float64)
    t = (t.round() if not dtype.is_floating_point else t).to(dtype)
    view = {1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[t.element_size()]
    return [int(v) for v in t.view(view).tolist()]


def _warp(img, coef8, kind, fill):
    squeeze = img.ndim == 2
    x = img.unsqueeze(0) if squeeze else img
    if not x.is_cuda:
        raise RuntimeError('semseg_amd: tensors must live on a HIP device (no CPU fallback)')
    coef = torch.tensor([coef8], dtype=torch.float32).to(x.device)
    out = ops.warp_nearest(x.contiguous(), coef, kind, _fill_bits(fill, x.dtype, x.shape[-3] if x.ndim >= 3 else 1))
    return out.squeeze(0) if squeeze else out


def affine(img, angle, translate, scale, shear, interpolation=NEAREST, fill=None, center=None):
    """torchvision.transforms.functional.affine on a [..., H, W] device tensor (translate in pixels, shear in degrees (x, y))."""
    if not _is_nearest(interpolation):
        raise NotImplementedError('semseg_amd.augment: only NEAREST resampling is built (what train.py uses)')
    h, w = img.shape[-2:]
    center_f = [0.0, 0.0] if center is None else [1.0 * (c - s * 0.5) for c, s in zip(center, [w, h])]
    m = inverse_affine_matrix(center_f, angle, [1.0 * t for t in translate], scale, shear)
    return _warp(img, m + [0.0, 0.0], 0, fill)


def perspective(img, startpoints, endpoints, interpolation=NEAREST, fill=None):
    if not _is_nearest(interpolation):
        raise NotImplementedError('semseg_amd.augment: only NEAREST resampling is built (what train.py uses)')
    return _warp(img, perspective_coeffs(startpoints, endpoints), 1, fill)


# ------------------------------------------------------------------ the reference's transform objects
def _pair(v, name, lo=None, hi=None, default=None):
    if v is None:
        return default
    if isinstance(v, (int, float)):
        v = (v, v)
    assert isinstance(v, Sequence) and len(v) == 2 and all(isinstance(t, (int, float)) for t in v), f"Invalid '{name}': {v}"
    assert all((lo is None or t >= lo) and (hi is None or t <= hi) for t in v), f"Invalid '{name}': {v}"
    return tuple(v)


class RandomAffine(object):
    """transforms.py:168-322: one draw of (angle, translate fractions, scale, shear) per call, applied identically to the image
    and -- always NEAREST -- to the label."""

    def __init__(self, degrees, translate=None, scale=None, shear=None, interpolation=NEAREST, fill=0, center=None):
        self.degrees = (degrees, degrees) if isinstance(degrees, (int, float)) else _pair(degrees, 'degrees', -180, 180)
        self.translate = _pair(translate, 'translate', 0, 1, (0, 0))
        self.scale = _pair(scale, 'scale', default=(1, 1))
        assert all(s > 0 for s in self.scale), f"Invalid 'scale': {scale}"
        if isinstance(shear, (int, float)):
            shear = (shear, shear, 0, 0)
        if shear is not None:
            assert isinstance(shear, Sequence) and len(shear) in (2, 4) and all(isinstance(s, (int, float)) and -180 <= s <= 180 for s in shear), \
                f"Invalid 'shear': {shear}"
            shear = tuple(shear) + (0, 0) if len(shear) == 2 else tuple(shear)
        self.shear = shear if shear is not None else (0, 0, 0, 0)
        self.interpolation = interpolation
        self.fill = 0 if fill is None else fill
        if center is not None:
            assert isinstance(center, Sequence) and len(center) == 2
        self.center = center

    @classmethod
    def affine_transform(cls, img, angle, translate, scale, shear, interpolation=NEAREST, fill=None, center=None):
        h, w = img.shape[-2:]
        translate = (translate[0] * h), (translate[1] * w)        # sic: the x fraction scales with the height (transforms.py:269)
        if center is not None:
            assert (0 < center[0] < h) and (0 < center[1] < w), f"'center'({center}) is outside of image"
        return affine(img, angle=angle, translate=translate, scale=scale, shear=shear, interpolation=interpolation, fill=fill, center=center)

    def __call__(self, *args, **kwargs):
        angle = random.uniform(self.degrees[0], self.degrees[1])
        translate = random.uniform(-self.translate[0], self.translate[0]), random.uniform(-self.translate[1], self.translate[1])
        scale = random.uniform(self.scale[0], self.scale[1])
        shear = random.uniform(self.shear[0], self.shear[1]), random.uniform(self.shear[2], self.shear[3])
        common = dict(angle=angle, translate=translate, scale=scale, shear=shear, fill=self.fill, center=self.center)
        yield (partial(RandomAffine.affine_transform, interpolation=self.interpolation, **common),
               partial(RandomAffine.affine_transform, interpolation=NEAREST, **common))


class RandomPerspective(object):
    """transforms.py:325-475: four corner end points drawn with torch.randint inside the distortion margins."""

    def __init__(self, image_shape, distortion_scale=0.5, interpolation=NEAREST, fill=None):
        if isinstance(image_shape, int):
            image_shape = (1, image_shape, image_shape)
        if isinstance(image_shape, Sequence) and len(image_shape) == 2:
            image_shape = (1,) + tuple(image_shape)
        assert isinstance(image_shape, Sequence) and len(image_shape) == 3 and all(isinstance(d, int) and d > 0 for d in image_shape), \
            f"Invalid 'image_shape': {image_shape}"
        assert isinstance(distortion_scale, (int, float)) and 0 <= distortion_scale <= 1, f"Invalid 'distortion_scale': {distortion_scale}"
        self.image_shape, self.distortion, self.interpolation = tuple(image_shape), distortion_scale, interpolation
        self.fill = 0 if fill is None else fill

    @classmethod
    def perspective_transform(cls, img, startpoints, endpoints, interpolation=NEAREST, fill=None):
        if fill is not None and not isinstance(fill, (int, float)) and img.ndim >= 2:
            channels = img.shape[-3] if img.ndim >= 3 else 1
            if len(fill) != channels and len(set(fill)) == 1:         # the label gets the image's per-channel tuple (transforms.py:470-475)
                fill = fill[0]
        return perspective(img, startpoints, endpoints, interpolation=interpolation, fill=fill)

    def __call__(self, *args, **kwargs):
        channels, height, width = self.image_shape
        dw, dh = int(self.distortion * (width // 2)), int(self.distortion * (height // 2))

        def draw(lo, hi):
            return int(torch.randint(lo, hi, size=(1,)).item())
        top_left = [draw(0, dw + 1), draw(0, dh + 1)]                 # same draws, same order as transforms.py:409-452
        top_right = [draw(width - dw - 1, width), draw(0, dh + 1)]
        bot_right = [draw(width - dw - 1, width), draw(height - dh - 1, height)]
        bot_left = [draw(0, dw + 1), draw(height - dh - 1, height)]
        start = [[0, 0], [width - 1, 0], [width - 1, height - 1], [0, height - 1]]
        end = [top_left, top_right, bot_right, bot_left]
        fill = (self.fill,) * channels if isinstance(self.fill, (int, float)) else self.fill
        yield (partial(RandomPerspective.perspective_transform, startpoints=start, endpoints=end, interpolation=self.interpolation, fill=fill),
               partial(RandomPerspective.perspective_transform, startpoints=start, endpoints=end, interpolation=NEAREST, fill=fill))


class SegmentationTransform(object):
    """transforms.py:40-147: a list of (transform, target, probability); each call yields one (image_fn, label_fn) pair -- mode
    'all' chains every entry that wins its toss, mode 'any' picks one entry."""
    _valid_modes = {'all', 'any'}
    _valid_targets = {'image', 'label', 'both', 'sync_pair'}

    def __init__(self, mode: str = 'all'):
        self._entries = []
        self.set_mode(mode.lower())

    def get_mode(self):
        return self._mode

    def set_mode(self, mode: str):
        if mode not in self._valid_modes:
            raise ValueError(f"Unsupported mode: {mode}!\nValid modes are: {self._valid_modes}")
        self._mode = mode

    def add_transform(self, transform: Callable, probability: float = 1, target: str = 'both'):
        target = target.lower()
        assert target in self._valid_targets, f"Invalid target ({target})!\nValid targets are: {self._valid_targets}"
        assert isinstance(transform, Callable), "specified transform is not callable"
        assert 0 <= probability <= 1, "Invalid probability!"
        self._entries.append((transform, target, probability))

    def __len__(self):
        return len(self._entries)

    @staticmethod
    def _split(transform, target, sync):
        ident = (lambda x: x)
        if target == 'both':
            return transform, transform
        if target == 'image':
            return transform, ident
        if target == 'label':
            return ident, transform
        return sync(transform)

    def __call__(self):
        ident = (lambda x: x)
        if self._mode == 'all':
            chain = []
            for transform, target, probability in self._entries:
                pair = (ident, ident)
                if toss() <= probability:
                    # the reference unpacks the generator object itself here (transforms.py:119-120), which only works for a generator
                    # that yields exactly two items; the pair the generator yields is what is meant
                    pair = self._split(transform, target, lambda t: next(t()))
                chain.append(pair)

            def run(x, k):
                for p in chain:
                    x = p[k](x)
                return x
            yield (lambda x: run(x, 0)), (lambda x: run(x, 1))
        else:
            transform, target, _ = choose(self._entries)
            yield self._split(transform, target, lambda t: next(t()))
