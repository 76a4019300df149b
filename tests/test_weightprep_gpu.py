"""weightprep: remembered weight conversions refreshed by one ss_weight_prep_batch launch must equal the single-tensor kernels."""
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


def _params():
    torch.manual_seed(0)
    shapes = [(64, 32), (320, 1280), (6, 512), (128, 1, 3, 3), (96, 1, 7, 7), (64, 32, 8, 8), (40, 3, 7, 7), (1000, 3)]
    return [nn.Parameter(torch.randn(s, device='cuda')) for s in shapes]


def _convert_all(ps, wp, dtype):
    out = []
    for p in ps:
        n = p.shape[0]
        out.append(wp.cast(p, dtype))
        out.append(wp.transpose(p, n, p.numel() // n, dtype))
        if p.dim() == 4 and p.shape[1] == 1:
            out.append(wp.transpose(p, n, p.numel() // n, torch.float32, reverse_cols=True))
        if p.dim() == 4 and p.shape[1] > 1:
            out.append(wp.pack(p, dtype))
            out.append(wp.pack(p, dtype, transposed=True))
            out.append(wp.pack(p, dtype, k_multiple=64))      # zero columns up to whole 64-column k-tiles
        if p.dim() == 2:
            out.append(wp.cast_padded(p, dtype, (n + 63) // 64 * 64))    # zero rows behind the real ones
    return out


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_batched_refresh_equals_single_tensor_kernels(dtype):
    from semseg_amd import weightprep as wp, ops
    wp.clear()
    ps = _params()
    first = _convert_all(ps, wp, dtype)                       # cold: single-tensor kernels, remembered
    again = _convert_all(ps, wp, dtype)
    assert all(a.data_ptr() == b.data_ptr() for a, b in zip(first, again))          # served from the remembered buffers
    with torch.no_grad():
        for p in ps:
            p.mul_(1.5).add_(0.25)                            # version counters move
    got = _convert_all(ps, wp, dtype)                         # first stale lookup replays everything in one launch
    assert all(a.data_ptr() == b.data_ptr() for a, b in zip(first, got))
    wp.clear()
    want = _convert_all(ps, wp, dtype)                        # fresh single-tensor conversions of the new values
    for g, w in zip(got, want):
        assert g.dtype == w.dtype and g.shape == w.shape and torch.equal(g, w)
    # the padded forms: real part = the plain conversion, padding exactly zero
    for p in ps:
        n = p.shape[0]
        if p.dim() == 2:
            full = wp.cast_padded(p, dtype, (n + 63) // 64 * 64)
            assert full.shape[0] % 64 == 0 and torch.equal(full[:n], wp.cast(p, dtype)) and float(full[n:].abs().sum()) == 0
        if p.dim() == 4 and p.shape[1] > 1:
            wide, base = wp.pack(p, dtype, k_multiple=64), wp.pack(p, dtype)
            assert wide.shape[1] % 64 == 0 and torch.equal(wide[:, :base.shape[1]], base) and float(wide[:, base.shape[1]:].abs().sum()) == 0
    # writes through .data do not move the version counter: bump() is the contract
    wp.cast(ps[0], torch.bfloat16)
    ps[0].data.zero_()
    assert not torch.equal(wp.cast(ps[0], torch.bfloat16).float(), torch.zeros_like(ps[0]))
    wp.bump()
    assert torch.equal(wp.cast(ps[0], torch.bfloat16).float(), torch.zeros_like(ps[0]))
    # tensors that are not parameters are converted on the spot and not remembered
    t = torch.randn(16, 8, device='cuda')
    a, b = wp.cast(t, torch.bfloat16), wp.cast(t, torch.bfloat16)
    assert a.data_ptr() != b.data_ptr() and torch.equal(a, ops.cast(t, torch.bfloat16))
    wp.clear()


def test_training_steps_see_updated_weights():
    """Two optimizer steps on a small MiT + Lawin: the loss must change exactly as it does with the cache cleared every step."""
    from semseg_amd import weightprep as wp
    from semseg_amd.model import SegFormerLawin
    from semseg_amd.loss import OhemCrossEntropyLoss
    from semseg_amd.engine import FlatParams, AdaBound
    from semseg_amd.detinit import det_fill_, det_input, det_labels
    from semseg_amd.runtime import set_compute_dtype
    x, lab = det_input('wp.x', (2, 3, 64, 64)).cuda(), det_labels('wp.lab', (2, 64, 64), 6, cell=8).cuda()
    losses = []
    set_compute_dtype(torch.bfloat16)
    try:
        for clear_each in (False, True):
            net = SegFormerLawin('B0', 128, 6)
            det_fill_(net.encoder), det_fill_(net.decoder)
            net = net.cuda().eval()
            flat = FlatParams(net)
            opt = AdaBound(flat, lr=1e-2)
            crit = OhemCrossEntropyLoss(0, 0.7, 0.1)
            run = []
            for _ in range(3):
                if clear_each:
                    wp.clear()
                flat.zero_grad()
                loss = crit(net(x), lab)
                loss.backward()
                opt.step()
                run.append(float(loss.detach()))
            losses.append(run)
    finally:
        set_compute_dtype(None)
    assert losses[0] == losses[1] and losses[0][0] != losses[0][2]
