"""Prediction writer (class maps) and evaluation statistics on the GPU vs the oracle (oracle/postproc.py) -- byte / integer
results, so bit-exact -- and the fused resize + arg-max vs the two kernels it replaces."""
import numpy as np
import pytest
import torch

from oracle import postproc as opost

pytestmark = pytest.mark.gpu


def _logits(shape, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * 3).to(dtype)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', [(2, 6, 33, 47), (1, 6, 1, 1), (3, 19, 16, 64), (1, 2, 5, 3)])
def test_class_map_vs_oracle(dtype, shape):
    from semseg_amd import postproc
    x = _logits(shape, dtype, 3)
    if shape[2] > 4:
        x[0, :, 2, 1] = 0.5                                   # full tie -> class 0
        x[0, shape[1] - 1, 3, 2] = float('nan')               # NaN pixel -> class 0
        x[0, 1, 4, 0] = x[0, shape[1] - 1, 4, 0] = 50.0       # two-way tie -> the first
    want = opost.class_map(x.float().numpy())
    got = postproc.class_map(x.cuda())                        # NCHW-contiguous logits (what the reference's models return)
    assert got.dtype == torch.uint8 and got.shape == (shape[0], shape[2], shape[3])
    assert np.array_equal(got.cpu().numpy(), want)
    cl = x.cuda().permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)      # channels-last storage (what this engine's models return)
    assert np.array_equal(postproc.class_map(cl).cpu().numpy(), want)
    if dtype == torch.float32 and shape[2] > 4:               # the reference expression itself, on the device
        ref = x.cuda().softmax(dim=1).argmax(dim=1)
        assert np.array_equal(ref.cpu().numpy().astype(np.uint8), want)


def test_prediction_writer_hands_over_the_reference_batch():
    from semseg_amd.postproc import PredictionWriter

    class DS:
        def write_batch(self, data_batch, indexes, overwrite):
            self.got = (data_batch, indexes, overwrite)

    ds = DS()
    x = _logits((2, 6, 40, 56), torch.float32, 9)
    ref = x.detach().clone().softmax(dim=1).argmax(dim=1, keepdim=True).cpu().numpy()          # helper/callbacks.py:52-54
    PredictionWriter(ds, overwrite=True).write_on_batch_end(None, None, x.cuda(), [4, 5], None, 0, 0)
    assert ds.got[0].dtype == np.uint8 and ds.got[0].shape == ref.shape and np.array_equal(ds.got[0], ref)
    assert ds.got[1] == [4, 5] and ds.got[2] is True
    PredictionWriter([ds], dtype=np.int64).write_on_batch_end(None, None, x.cuda(), [0, 1], None, 0, 0)
    assert ds.got[0].dtype == ref.dtype and np.array_equal(ds.got[0], ref)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('geom', [(2, 16, 16, 64, 64, 6, 8), (1, 13, 9, 50, 35, 6, 8), (2, 8, 8, 33, 1027, 3, 8), (1, 5, 7, 5, 7, 19, 24)])
def test_fused_resize_class_map_equals_the_two_kernels(dtype, geom):
    from semseg_amd import ops
    B, hi, wi, ho, wo, C, ld = geom
    x = _logits((B * hi * wi, ld), dtype, 21).cuda()
    x[:, C:] = 100.0                                          # padding columns must not be looked at
    full = ops.bilinear_fwd(x[:, :C].contiguous() if C % 8 else x, B, hi, wi, ho, wo)
    want = ops.class_map(full.view(B, ho, wo, -1)[..., :C].permute(0, 3, 1, 2))
    got = ops.bilinear_class_map(x, B, hi, wi, ho, wo, C)
    assert torch.equal(got, want)
    assert np.array_equal(want.cpu().numpy(), opost.class_map(full.view(B, ho, wo, -1)[..., :C].permute(0, 3, 1, 2).float().cpu().numpy()))


@pytest.mark.parametrize('head', ['simple', 'refiner'])
def test_topformer_predict_class_map_equals_forward_then_writer(head):
    from semseg_amd import postproc
    from semseg_amd.topformer import TopFormerModel
    from semseg_amd.detinit import det_fill_, det_input
    m = det_fill_(TopFormerModel(6, 'T', head_cfg={'alias': head}))
    g = torch.Generator().manual_seed(4)
    for name, p in m.classifier.named_parameters():              # a classifier whose winner changes from pixel to pixel
        p.data = torch.randn(p.shape, generator=g) if p.dim() > 1 else torch.zeros_like(p)
    m = m.cuda().eval()
    x = det_input('pp.x', (2, 3, 256, 192), 0, 1).cuda()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        want = postproc.class_map(m(x))
        got = m.predict_class_map(x)
    assert got.shape == (2, 256, 192) and torch.equal(got, want)
    assert len(torch.unique(got)) > 2 and not torch.isnan(m(x).float()).any()


def test_image_statistics_vs_oracle():
    from semseg_amd import postproc
    rng = np.random.default_rng(2)
    labels = [0, 1, 2, 3, 4, 5]
    for case in range(4):
        gt = rng.integers(0, 6, size=(301, 203)).astype(np.uint8)
        pr = np.where(rng.random(gt.shape) < 0.8, gt, rng.integers(0, 6, size=gt.shape)).astype(np.uint8)
        if case == 1:
            gt[gt == 4] = 1; pr[pr == 4] = 1; pr[pr == 5] = 0
        if case == 2:
            gt[:4] = 9; pr[-3:] = 200
        valid = (rng.random(gt.shape) < 0.9) if case == 3 else None
        want = opost.image_statistics(opost.confusion(gt, pr, labels, valid))
        got = postproc.image_statistics(torch.from_numpy(gt).cuda(), torch.from_numpy(pr).cuda(), labels,
                                        None if valid is None else torch.from_numpy(valid).cuda())
        assert np.array_equal(got['cm'], want['cm'])                       # integer counts: exact
        for k in ('f1', 'iou', 'norm_cm', 'accuracy'):
            np.testing.assert_allclose(got[k], want[k], rtol=0, atol=1e-15)
        assert abs(got['kappa'] - want['kappa']) < 1e-15
    sub = postproc.image_statistics(torch.from_numpy(gt).cuda(), torch.from_numpy(pr).cuda(), [1, 3, 5])   # a subset of the classes, reordered rows
    assert np.array_equal(sub['cm'], opost.confusion(gt, pr, [1, 3, 5])[:3, :3])


def test_full_size_class_map_properties():
    """BASELINE configs[3] size (4 x 6 x 2048 x 2048, bf16): every pixel gets a class, a logit shift leaves the map unchanged,
    and the class histogram equals the column sums of the confusion matrix against any truth raster."""
    from semseg_amd import ops, postproc
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn((4, 2048, 2048, 6), generator=g, device='cuda').to(torch.bfloat16).permute(0, 3, 1, 2)
    cm = postproc.class_map(x)
    hist = torch.bincount(cm.reshape(-1).long(), minlength=6)
    assert int(hist.sum()) == 4 * 2048 * 2048 and int(cm.max()) <= 5
    assert torch.equal(cm, x.float().argmax(dim=1).to(torch.uint8))
    assert torch.equal(postproc.class_map(x.float() * 2.0 + 1.0), cm)
    truth = torch.randint(0, 6, cm.shape, generator=g, device='cuda', dtype=torch.uint8)
    full = postproc.confusion_matrix(truth, cm, list(range(6)))
    assert torch.equal(full[:6, :6].sum(0), hist) and int(full[6].sum()) == 0 and int(full.sum()) == cm.numel()
