"""uint8 input staging on the GPU vs the reference expressions of data_factory/utils.py:6-37 evaluated by torch on the CPU
(x - mu) / sigma in float32 is one subtraction and one correctly rounded division per value, so the fp32 result is bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def reference_image_to_tensor(arr, mu=None, sigma=None):
    """data_factory/utils.py:6-33 restated line by line (test infrastructure)."""
    c = arr.shape[0]
    x = torch.from_numpy(arr).contiguous()
    mu = [0] * c if mu is None else mu
    sigma = [255] * c if sigma is None else sigma
    mu = torch.tensor(data=np.expand_dims(a=np.array(mu), axis=(1, 2)), dtype=torch.float32)
    sigma = torch.tensor(data=np.expand_dims(a=np.array(sigma), axis=(1, 2)), dtype=torch.float32)
    return (x - mu) / sigma


@pytest.mark.parametrize('shape', [(3, 64, 64), (3, 37, 53), (4, 5, 3), (1, 1, 1)])
@pytest.mark.parametrize('stats', [(None, None), ([123.675, 116.28, 103.53, 90.0], [58.395, 57.12, 57.375, 61.0])])
def test_image_to_tensor_fp32_is_the_reference_tensor(shape, stats):
    from semseg_amd.datastage import image_to_tensor
    rng = np.random.default_rng(7)
    arr = rng.integers(0, 256, size=shape, dtype=np.uint8)
    mu, sigma = (None, None) if stats[0] is None else (stats[0][:shape[0]], stats[1][:shape[0]])
    want = reference_image_to_tensor(arr, mu, sigma)
    got = image_to_tensor(torch.from_numpy(arr).cuda(), mu, sigma, dtype=torch.float32)
    assert got.dtype == torch.float32 and got.is_contiguous() and torch.equal(got.cpu(), want)
    batch = np.stack([arr, arr[:, ::-1].copy()])
    gb = image_to_tensor(torch.from_numpy(batch).cuda(), mu, sigma, dtype=torch.float32)
    assert torch.equal(gb[0].cpu(), want) and torch.equal(gb[1].cpu(), reference_image_to_tensor(batch[1], mu, sigma))


def test_image_to_tensor_compute_dtype_is_channels_last_and_feeds_the_model_without_a_copy():
    from semseg_amd.datastage import image_to_tensor, label_to_tensor
    from semseg_amd.runtime import as_tokens_nhwc
    from semseg_amd.segformer import SegFormer
    from semseg_amd.detinit import det_fill_
    rng = np.random.default_rng(3)
    batch = rng.integers(0, 256, size=(2, 3, 64, 96), dtype=np.uint8)
    ref = torch.stack([reference_image_to_tensor(b) for b in batch])
    with torch.autocast('cuda', dtype=torch.bfloat16):
        x = image_to_tensor(torch.from_numpy(batch).cuda())
        assert x.dtype == torch.bfloat16 and x.shape == (2, 3, 64, 96) and x.permute(0, 2, 3, 1).is_contiguous()
        assert torch.equal(x.cpu(), ref.to(torch.bfloat16))
        assert as_tokens_nhwc(x, torch.bfloat16).data_ptr() == x.data_ptr()
        net = det_fill_(SegFormer('B0')).cuda().eval()
        with torch.no_grad():
            a, b = net(x), net(ref.cuda())
        assert all(torch.equal(p, q) for p, q in zip(a, b))
    lab = rng.integers(0, 6, size=(2, 64, 96), dtype=np.uint8)
    got = label_to_tensor(torch.from_numpy(lab).cuda())
    assert got.dtype == torch.int64 and torch.equal(got.cpu(), torch.from_numpy(lab).contiguous().to(dtype=torch.int64))


def test_staging_refuses_cpu_tensors_and_bad_sigma():
    from semseg_amd.datastage import image_to_tensor
    with pytest.raises(RuntimeError):
        image_to_tensor(torch.zeros((3, 4, 4), dtype=torch.uint8))
    with pytest.raises(AssertionError):
        image_to_tensor(torch.zeros((3, 4, 4), dtype=torch.uint8).cuda(), [0, 0, 0], [1, 0, 1])
