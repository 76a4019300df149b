"""CPU oracle: geometric augmentation.  TEST INFRASTRUCTURE ONLY (imported by tests/ alone).

Parity unpinned: the reference (data_factory/transforms.py:168-475, train.py:69-113) delegates to
torchvision.transforms.functional.{hflip, vflip, affine, perspective}, and torchvision is neither vendored in /root/reference nor
installed here; the reference holds no fixture for these.  Two restatements of torchvision's published tensor algorithm:

* ``warp_grid_sample`` -- the literal pipeline: float32 base grid, bmm with the rescaled matrix, F.grid_sample(nearest, zeros,
  align_corners=False), mask channel -> fill.  Its bmm rounds in an implementation-defined order.
* ``warp_exact`` -- the same arithmetic with every float32 operation written out in one fixed order (numpy), the order the HIP
  kernel uses: the bit-exact checker.  The two agree except on pixels whose source lies within an ulp of a rounding boundary.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F


def inverse_affine_matrix(center, angle, translate, scale, shear):
    rot, sx, sy = math.radians(angle), math.radians(shear[0]), math.radians(shear[1])
    cx, cy = center
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]
    m = [v / scale for v in m]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return m


def perspective_coeffs(startpoints, endpoints):
    a = torch.zeros(8, 8, dtype=torch.float64)
    for i, (p1, p2) in enumerate(zip(endpoints, startpoints)):
        a[2 * i, :] = torch.tensor([p1[0], p1[1], 1, 0, 0, 0, -p2[0] * p1[0], -p2[0] * p1[1]], dtype=torch.float64)
        a[2 * i + 1, :] = torch.tensor([0, 0, 0, p1[0], p1[1], 1, -p2[1] * p1[0], -p2[1] * p1[1]], dtype=torch.float64)
    b = torch.tensor(startpoints, dtype=torch.float64).view(8)
    return torch.linalg.lstsq(a, b, driver='gels').solution.to(torch.float32).tolist()


def _grid(kind, coef, h, w):
    if kind == 0:
        theta = torch.tensor(coef[:6], dtype=torch.float32).reshape(1, 2, 3)
        base = torch.empty(1, h, w, 3, dtype=torch.float32)
        base[..., 0].copy_(torch.linspace(-w * 0.5 + 0.5, w * 0.5 + 0.5 - 1, steps=w))
        base[..., 1].copy_(torch.linspace(-h * 0.5 + 0.5, h * 0.5 + 0.5 - 1, steps=h).unsqueeze_(-1))
        base[..., 2].fill_(1)
        rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h], dtype=torch.float32)
        return base.view(1, h * w, 3).bmm(rescaled).view(1, h, w, 2)
    theta1 = torch.tensor([[[coef[0], coef[1], coef[2]], [coef[3], coef[4], coef[5]]]], dtype=torch.float32)
    theta2 = torch.tensor([[[coef[6], coef[7], 1.0], [coef[6], coef[7], 1.0]]], dtype=torch.float32)
    base = torch.empty(1, h, w, 3, dtype=torch.float32)
    base[..., 0].copy_(torch.linspace(0.5, w * 1.0 + 0.5 - 1.0, steps=w))
    base[..., 1].copy_(torch.linspace(0.5, h * 1.0 + 0.5 - 1.0, steps=h).unsqueeze_(-1))
    base[..., 2].fill_(1)
    g1 = base.view(1, h * w, 3).bmm(theta1.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h], dtype=torch.float32))
    g2 = base.view(1, h * w, 3).bmm(theta2.transpose(1, 2))
    return (g1 / g2 - 1.0).view(1, h, w, 2)


def warp_grid_sample(img, kind, coef, fill):
    """img [C, H, W] (any dtype) -> torchvision's _apply_grid_transform with mode nearest."""
    c, h, w = img.shape
    x = img.to(torch.float32).unsqueeze(0)
    x = torch.cat((x, torch.ones(1, 1, h, w)), dim=1)
    y = F.grid_sample(x, _grid(kind, coef, h, w), mode='nearest', padding_mode='zeros', align_corners=False)
    mask, y = y[:, -1:], y[:, :-1]
    fv = [float(fill)] * c if isinstance(fill, (int, float)) else [float(v) for v in fill]
    fill_img = torch.tensor(fv, dtype=torch.float32).view(1, -1, 1, 1).expand_as(y)
    y = torch.where((mask < 0.5).expand_as(y), fill_img, y)[0]
    return y.round().to(img.dtype) if not img.dtype.is_floating_point else y.to(img.dtype)


def source_index_exact(kind, coef, h, w):
    """-> (iy, ix, inside) int64 / bool [h, w]: the fixed-order float32 evaluation (matches csrc/augment.hip warp_source)."""
    f = np.float32
    c = [f(v) for v in coef]
    xs, ys = np.arange(w, dtype=np.float32)[None, :], np.arange(h, dtype=np.float32)[:, None]
    hw, hh = f(0.5) * f(w), f(0.5) * f(h)
    if kind == 0:
        bx, by = xs + (f(0.5) - f(0.5) * f(w)), ys + (f(0.5) - f(0.5) * f(h))
        gx = (bx * (c[0] / hw) + by * (c[1] / hw)) + (c[2] / hw)
        gy = (bx * (c[3] / hh) + by * (c[4] / hh)) + (c[5] / hh)
    else:
        bx, by = xs + f(0.5), ys + f(0.5)
        nx = (bx * (c[0] / hw) + by * (c[1] / hw)) + (c[2] / hw)
        ny = (bx * (c[3] / hh) + by * (c[4] / hh)) + (c[5] / hh)
        with np.errstate(divide='ignore', invalid='ignore'):
            den = (bx * c[6] + by * c[7]) + f(1.0)
            gx, gy = nx / den + f(-1.0), ny / den + f(-1.0)
    assert gx.dtype == np.float32 and gy.dtype == np.float32
    sx = ((gx + f(1.0)) * f(w) + f(-1.0)) / f(2.0)
    sy = ((gy + f(1.0)) * f(h) + f(-1.0)) / f(2.0)
    rx, ry = np.rint(sx), np.rint(sy)
    inside = (rx >= 0) & (rx < w) & (ry >= 0) & (ry < h)
    ix = np.where(inside, rx, 0).astype(np.int64)
    iy = np.where(inside, ry, 0).astype(np.int64)
    return torch.from_numpy(iy), torch.from_numpy(ix), torch.from_numpy(inside)


def warp_exact(img, kind, coef, fill):
    c, h, w = img.shape
    iy, ix, inside = source_index_exact(kind, coef, h, w)
    fv = [fill] * c if isinstance(fill, (int, float)) else list(fill)
    ft = torch.tensor(fv, dtype=torch.float64)
    ft = (ft.round() if not img.dtype.is_floating_point else ft).to(img.dtype).view(c, 1, 1)
    return torch.where(inside.unsqueeze(0), img[:, iy, ix], ft.expand(c, h, w))
