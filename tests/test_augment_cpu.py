"""Host logic of semseg_amd.augment and the augmentation oracle (no GPU): parameter math, random draws, the two oracle forms."""
import math
import random

import numpy as np
import pytest
import torch

from oracle import augment as oaug
from semseg_amd import augment


def test_inverse_affine_matrix_matches_oracle_and_closed_forms():
    for args in [((0.0, 0.0), 0.0, (0.0, 0.0), 1.0, (0.0, 0.0)), ((3.5, -2.0), 33.0, (4.0, -7.5), 1.2, (10.0, -5.0)),
                 ((0.0, 0.0), -170.0, (0.0, 12.0), 0.8, (0.0, 0.0))]:
        a, b = augment.inverse_affine_matrix(*args), oaug.inverse_affine_matrix(*args)
        assert np.allclose(a, b, rtol=1e-12, atol=1e-12)
    assert np.allclose(augment.inverse_affine_matrix((0, 0), 0, (0, 0), 1, (0, 0)), [1, 0, 0, 0, 1, 0])
    # pure translation by (tx, ty): the output pixel reads the input at (x - tx, y - ty)
    assert np.allclose(augment.inverse_affine_matrix((0, 0), 0, (5, -3), 1, (0, 0)), [1, 0, -5, 0, 1, 3])
    # rotation + scale: inverse of scale * R(angle)
    m = augment.inverse_affine_matrix((0, 0), 30, (0, 0), 2, (0, 0))
    fwd = 2 * np.array([[math.cos(math.radians(30)), -math.sin(math.radians(30))], [math.sin(math.radians(30)), math.cos(math.radians(30))]])
    assert np.allclose(np.array([[m[0], m[1]], [m[3], m[4]]]) @ fwd, np.eye(2), atol=1e-12)


def test_perspective_coeffs_map_endpoints_to_startpoints():
    start = [[0, 0], [63, 0], [63, 47], [0, 47]]
    end = [[3, 2], [60, 4], [58, 45], [5, 40]]
    c = augment.perspective_coeffs(start, end)
    assert np.allclose(c, oaug.perspective_coeffs(start, end), rtol=1e-4, atol=1e-5)
    for (ex, ey), (sx, sy) in zip(end, start):
        den = c[6] * ex + c[7] * ey + 1
        assert abs((c[0] * ex + c[1] * ey + c[2]) / den - sx) < 1e-6 and abs((c[3] * ex + c[4] * ey + c[5]) / den - sy) < 1e-6


@pytest.mark.parametrize('kind', [0, 1])
def test_oracle_exact_form_agrees_with_grid_sample_form(kind):
    g = torch.Generator().manual_seed(kind)
    img = torch.randint(0, 255, (3, 96, 80), generator=g, dtype=torch.uint8)
    if kind == 0:
        coef = oaug.inverse_affine_matrix((0.0, 0.0), 37.0, (6.0, -9.0), 0.9, (8.0, 0.0)) + [0.0, 0.0]
    else:
        coef = oaug.perspective_coeffs([[0, 0], [79, 0], [79, 95], [0, 95]], [[4, 3], [75, 6], [70, 90], [7, 88]])
    a, b = oaug.warp_exact(img, kind, coef, (7, 8, 9)), oaug.warp_grid_sample(img, kind, coef, (7, 8, 9))
    assert float((a != b).float().mean()) < 2e-3          # rounding-boundary pixels only
    ident = oaug.warp_exact(img, 0, [1, 0, 0, 0, 1, 0, 0, 0], 0)
    assert torch.equal(ident, img)


def test_random_affine_draws_follow_the_reference_order():
    aug = augment.RandomAffine(degrees=(-180, 180), translate=(0.15, 0.15), scale=(0.8, 1.2), shear=None, fill=0)
    random.seed(11)
    img_fn, lab_fn = next(aug())
    random.seed(11)
    angle = random.uniform(-180, 180)
    tr = random.uniform(-0.15, 0.15), random.uniform(-0.15, 0.15)
    sc = random.uniform(0.8, 1.2)
    sh = random.uniform(0, 0), random.uniform(0, 0)
    for fn in (img_fn, lab_fn):
        assert fn.keywords['angle'] == angle and fn.keywords['translate'] == tr and fn.keywords['scale'] == sc and fn.keywords['shear'] == sh
    assert augment.RandomAffine(30).degrees == (30, 30) and augment.RandomAffine(0, shear=(1, 2)).shear == (1, 2, 0, 0)
    with pytest.raises(AssertionError):
        augment.RandomAffine(degrees=(0, 270))


def test_random_perspective_draws_follow_the_reference_order():
    aug = augment.RandomPerspective(image_shape=(3, 64, 48), distortion_scale=0.2, fill=0)
    torch.manual_seed(5)
    img_fn, lab_fn = next(aug())
    torch.manual_seed(5)
    dw, dh = int(0.2 * 24), int(0.2 * 32)
    r = lambda lo, hi: int(torch.randint(lo, hi, size=(1,)).item())   # noqa: E731
    tl = [r(0, dw + 1), r(0, dh + 1)]
    tr = [r(48 - dw - 1, 48), r(0, dh + 1)]
    br = [r(48 - dw - 1, 48), r(64 - dh - 1, 64)]
    bl = [r(0, dw + 1), r(64 - dh - 1, 64)]
    assert img_fn.keywords['endpoints'] == [tl, tr, br, bl] and lab_fn.keywords['endpoints'] == [tl, tr, br, bl]
    assert img_fn.keywords['startpoints'] == [[0, 0], [47, 0], [47, 63], [0, 63]] and img_fn.keywords['fill'] == (0, 0, 0)


def test_segmentation_transform_scheduling():
    t = augment.SegmentationTransform(mode='ALL'.lower())
    t.add_transform(lambda x: x + 1, probability=1, target='both')
    t.add_transform(lambda x: x * 10, probability=1, target='image')
    t.add_transform(lambda x: x - 5, probability=0, target='label')
    fi, fl = next(t())
    assert fi(1) == 20 and fl(1) == 2 and len(t) == 3

    def sync():
        yield (lambda x: ('img', x)), (lambda x: ('lab', x))
    t2 = augment.SegmentationTransform(mode='any')
    t2.add_transform(sync, target='sync_pair')
    fi, fl = next(t2())
    assert fi(3) == ('img', 3) and fl(3) == ('lab', 3)
    with pytest.raises(ValueError):
        augment.SegmentationTransform(mode='some')
    with pytest.raises(AssertionError):
        t2.add_transform(sync, target='nobody')


def test_device_transforms_refuse_cpu_tensors():
    with pytest.raises(RuntimeError):
        augment.affine(torch.zeros(3, 8, 8), 10.0, (0, 0), 1.0, (0.0, 0.0))
    with pytest.raises(NotImplementedError):
        augment.affine(torch.zeros(3, 8, 8), 10.0, (0, 0), 1.0, (0.0, 0.0), interpolation='bilinear')
