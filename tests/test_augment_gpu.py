"""semseg_amd.augment on the GPU (ss_flip / ss_warp_nearest) against oracle/augment.py: bit-exact vs the fixed-order restatement of
torchvision's nearest affine / perspective, within rounding-boundary pixels of the literal grid_sample pipeline.  Parity unpinned
(torchvision is not vendored in the reference; no fixture exists)."""
import random

import pytest
import torch

from oracle import augment as oaug
from semseg_amd import augment, ops

pytestmark = pytest.mark.gpu


def _img(dtype, shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    if dtype == torch.uint8:
        return torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
    if dtype == torch.int64:
        return torch.randint(0, 10, shape, generator=g, dtype=torch.int64)
    return torch.randn(shape, generator=g).to(dtype)


@pytest.mark.parametrize('dtype', [torch.uint8, torch.int64, torch.float32, torch.bfloat16])
def test_flips_are_exact(dtype):
    x = _img(dtype, (2, 3, 37, 300))
    assert torch.equal(augment.hflip(x.cuda()).cpu(), torch.flip(x, dims=[-1]))
    assert torch.equal(augment.vflip(x.cuda()).cpu(), torch.flip(x, dims=[-2]))
    lab = _img(torch.int64, (45, 31))
    assert torch.equal(augment.hflip(lab.cuda()).cpu(), torch.flip(lab, dims=[-1]))
    assert torch.equal(augment.vflip(augment.vflip(lab.cuda())).cpu(), lab)


AFFINE_CASES = [dict(angle=0.0, translate=(0, 0), scale=1.0, shear=(0.0, 0.0)),
                dict(angle=33.0, translate=(7, -12), scale=1.15, shear=(0.0, 0.0)),
                dict(angle=-171.5, translate=(-20, 3), scale=0.8, shear=(12.0, -7.0)),
                dict(angle=90.0, translate=(0, 0), scale=1.0, shear=(0.0, 0.0), center=(40, 30))]


@pytest.mark.parametrize('dtype', [torch.uint8, torch.float32, torch.bfloat16])
@pytest.mark.parametrize('case', range(len(AFFINE_CASES)))
def test_affine_matches_oracle(dtype, case):
    kw = dict(AFFINE_CASES[case])
    img = _img(dtype, (3, 120, 88), seed=case)
    fill = (3, 250, 17)
    out = augment.affine(img.cuda(), fill=fill, **kw).cpu()
    h, w = img.shape[-2:]
    center = kw.get('center')
    cf = [0.0, 0.0] if center is None else [1.0 * (c - s * 0.5) for c, s in zip(center, [w, h])]
    coef = oaug.inverse_affine_matrix(cf, kw['angle'], [1.0 * t for t in kw['translate']], kw['scale'], kw['shear']) + [0.0, 0.0]
    assert torch.equal(out, oaug.warp_exact(img, 0, coef, fill))
    lit = oaug.warp_grid_sample(img, 0, coef, fill)
    assert float((out != lit).float().mean()) < 2e-3
    if case == 0:
        assert torch.equal(out, img)


def test_affine_label_and_reference_translate_quirk():
    """labels are [H, W] int64 (the reference un-squeezes them); RandomAffine.affine_transform scales the x fraction by the HEIGHT."""
    lab = _img(torch.int64, (64, 96))
    out = augment.RandomAffine.affine_transform(lab.cuda(), angle=15.0, translate=(0.1, -0.05), scale=0.9, shear=(0.0, 0.0), fill=0).cpu()
    coef = oaug.inverse_affine_matrix([0.0, 0.0], 15.0, [0.1 * 64, -0.05 * 96], 0.9, (0.0, 0.0)) + [0.0, 0.0]
    assert out.shape == lab.shape and torch.equal(out, oaug.warp_exact(lab.unsqueeze(0), 0, coef, 0)[0])


@pytest.mark.parametrize('dtype', [torch.uint8, torch.int64, torch.float32])
def test_perspective_matches_oracle(dtype):
    img = _img(dtype, (3, 100, 140), seed=3)
    start = [[0, 0], [139, 0], [139, 99], [0, 99]]
    end = [[9, 4], [131, 7], [135, 92], [3, 95]]
    fill = (0, 5, 9)
    out = augment.perspective(img.cuda(), start, end, fill=fill).cpu()
    coef = augment.perspective_coeffs(start, end)
    assert torch.equal(out, oaug.warp_exact(img, 1, coef, fill))
    lit = oaug.warp_grid_sample(img, 1, oaug.perspective_coeffs(start, end), fill)
    assert float((out != lit).float().mean()) < 5e-3
    same = augment.perspective(img.cuda(), start, start, fill=fill).cpu()
    assert float((same != img).float().mean()) < 1e-3        # identity up to the least-squares rounding of the coefficients


def test_one_launch_per_batch_with_a_transform_per_sample():
    x = _img(torch.uint8, (4, 3, 50, 70), seed=9)
    coefs = [oaug.inverse_affine_matrix((0.0, 0.0), a, (t, -t), s, (0.0, 0.0)) + [0.0, 0.0] for a, t, s in ((10.0, 3.0, 1.0), (-45.0, 0.0, 1.2), (120.0, 8.0, 0.85), (0.0, 0.0, 1.0))]
    out = ops.warp_nearest(x.cuda(), torch.tensor(coefs, dtype=torch.float32).cuda(), 0, [1, 2, 3]).cpu()
    for b in range(4):
        assert torch.equal(out[b], oaug.warp_exact(x[b], 0, coefs[b], (1, 2, 3))), b
    with pytest.raises(ValueError):
        ops.warp_nearest(x.cuda(), torch.tensor(coefs + coefs[:1], dtype=torch.float32).cuda(), 0)      # 12 planes, 5 transforms


def test_training_augmentor_as_in_train_py():
    """train.py:68-113: mode 'any' over hflip / vflip / RandomAffine / RandomPerspective, image and label transformed in sync."""
    h = w = 128
    aug = augment.SegmentationTransform(mode='any')
    aug.add_transform(augment.hflip, target='both', probability=0.5)
    aug.add_transform(augment.vflip, target='both', probability=0.5)
    aug.add_transform(augment.RandomAffine(degrees=(-180, 180), translate=(0.15, 0.15), scale=(0.8, 1.2), shear=None, fill=0), target='sync_pair', probability=0.5)
    aug.add_transform(augment.RandomPerspective(image_shape=(3, h, w), distortion_scale=0.2, fill=0), target='sync_pair', probability=0.5)
    lab = torch.randint(1, 6, (h // 16, w // 16)).repeat_interleave(16, 0).repeat_interleave(16, 1)
    img = torch.stack([lab * 40, lab * 20 + 3, 255 - lab]).to(torch.uint8)          # every channel is a function of the label
    random.seed(4), torch.manual_seed(4)
    seen = set()
    for _ in range(12):
        fi, fl = next(aug())
        i2, l2 = fi(img.cuda()).cpu(), fl(lab.cuda()).cpu()
        assert i2.shape == img.shape and l2.shape == lab.shape and i2.dtype == torch.uint8 and l2.dtype == torch.int64
        inside = l2 > 0                                                             # fill 0 marks the pixels that left the tile
        assert torch.equal(i2[0][inside], (l2[inside] * 40).to(torch.uint8)) and torch.equal(i2[2][inside], (255 - l2[inside]).to(torch.uint8))
        assert bool((i2[:, ~inside] == 0).all())
        seen.add(getattr(getattr(fi, 'func', fi), '__name__', 'fn'))
    assert len(seen) >= 2


def test_full_size_tile_properties():
    """BASELINE tile size: identity, flip involution and a 180 degree turn equal to both flips, on a 1024 x 1024 label."""
    lab = torch.randint(0, 6, (1024, 1024), dtype=torch.int64).cuda()
    assert torch.equal(augment.affine(lab, 0.0, (0, 0), 1.0, (0.0, 0.0)), lab)
    assert torch.equal(augment.hflip(augment.hflip(lab)), lab)
    assert torch.equal(augment.affine(lab, 180.0, (0, 0), 1.0, (0.0, 0.0)), augment.vflip(augment.hflip(lab)))
