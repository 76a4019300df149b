"""MiT encoder on the GPU (HIP path through the C ABI) vs the golden vectors captured from the reference and vs the
CPU oracle on the same seeded inputs.  fp32: 1e-4 relative on outputs AND gradients; bf16: 2e-2 (relative L2) on outputs, gradients within max(2e-2, 1.5 x the reference's
own bf16-autocast deviation).  Measured values: gpurun_out/parity_errors.json, DESIGN.md section 5."""
import os
import numpy as np
import pytest
import torch

from semseg_amd.detinit import det_fill_, det_input, det_state_dict
from semseg_amd import shapes

pytestmark = pytest.mark.gpu


from _parity import rel, rel_l2  # noqa: E402,F401  (recording error measures)



def _run(dtype, x, projs_dtype=None):
    from semseg_amd.segformer import SegFormer
    from semseg_amd.runtime import set_compute_dtype
    enc = det_fill_(SegFormer('B0')).cuda().eval()
    set_compute_dtype(dtype)
    try:
        outs = enc(x.cuda())
        projs = [det_input(f'mit.proj{i}', o.shape, -1, 1).cuda() for i, o in enumerate(outs)]
        sum((o.float() * p).sum() for o, p in zip(outs, projs)).backward()
    finally:
        set_compute_dtype(None)
    return enc, outs


@pytest.mark.parametrize('fused_dw', [False, True])
def test_mit_b0_fp32_matches_reference_golden(golden_dir, monkeypatch, fused_dw):
    from semseg_amd import segformer as sf
    monkeypatch.setattr(sf, 'FUSED_DW_MIN_PIXELS', 0 if fused_dw else 1 << 30)     # both depthwise-backward forms on the 16 x 16 .. 2 x 2 maps
    g = np.load(os.path.join(golden_dir, 'mit_b0_64.npz'))
    enc, outs = _run(torch.float32, torch.from_numpy(g['x']))
    for i in range(4):
        assert outs[i].shape == g[f'out{i}'].shape
        assert rel(outs[i], g[f'out{i}']) < 1e-4, i
    grads = dict(enc.named_parameters())
    for k in g.files:
        if k.startswith('grad.'):
            assert rel(grads[k[5:]].grad, g[k]) < 1e-4, k


@pytest.mark.parametrize('fused_ffn', [False, True])
@pytest.mark.parametrize('fused_dw', [False, True])
def test_mit_b0_bf16_matches_reference_golden(golden_dir, monkeypatch, fused_dw, fused_ffn):
    from semseg_amd import segformer as sf
    monkeypatch.setattr(sf, 'FUSED_DW_MIN_PIXELS', 0 if fused_dw else 1 << 30)
    monkeypatch.setattr(sf, 'FUSED_MIXFFN', fused_ffn)                              # stage 2 of MiT-B0 (C = 64) takes the one-kernel Mix-FFN
    g = np.load(os.path.join(golden_dir, 'mit_b0_64.npz'))
    enc, outs = _run(torch.bfloat16, torch.from_numpy(g['x']))
    for i in range(4):
        assert outs[i].dtype == torch.bfloat16
        assert rel_l2(outs[i], g[f'out{i}']) < 2e-2, i
    grads = dict(enc.named_parameters())
    for k in g.files:
        if k.startswith('grad.'):
            assert grads[k[5:]].grad.dtype == torch.float32
            # 2e-2, or 1.5 x the deviation of the reference's OWN bf16-autocast gradient from its fp32 one where that is larger
            # (bf16.* vectors of the fixture: 0.5e-2 .. 1.6e-2 here; the reference keeps its residual stream in fp32 under autocast)
            assert rel_l2(grads[k[5:]].grad, g[k]) < max(2e-2, 1.5 * rel_l2(g['bf16.' + k], g[k])), k


def test_mit_b0_training_droppath_vs_oracle(monkeypatch):
    """train mode: DropPath multipliers fused into the GEMM epilogues; non-square 96x64 input (ragged Nk = 3*2)."""
    from oracle import mit as omit
    from semseg_amd import segformer as sf
    gen = torch.Generator().manual_seed(7)
    keeps = []

    def fake_keep(b, dpr, dtype, device):
        k = (torch.rand(b, generator=gen) > 0.3).float() / 0.7
        keeps.append(k)
        return k.to(device)

    monkeypatch.setattr(sf, 'drop_path_keep', fake_keep)
    enc = det_fill_(sf.SegFormer('B0')).cuda().train()
    x = det_input('mit.x2', (2, 3, 96, 64))
    outs = enc(x.cuda())
    projs = [det_input(f'mit.p2{i}', o.shape, -1, 1) for i, o in enumerate(outs)]
    sum((o * p.cuda()).sum() for o, p in zip(outs, projs)).backward()
    # oracle with the same multipliers (block1.0 has dpr 0 -> no draw, modules/segformer.py:92)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in det_state_dict(shapes.mit_shapes('B0', 3)).items()}
    it = iter(keeps)
    kmap = {}
    for s in range(4):
        for i in range(2):
            if (s, i) != (0, 0):
                kmap[(s, i)] = (next(it), next(it))
    oouts = omit.mit_forward(sd, x, 'B0', keeps=kmap)
    sum((o * p).sum() for o, p in zip(oouts, projs)).backward()
    for i in range(4):
        assert rel(outs[i], oouts[i]) < 1e-4
    for k, p in enc.named_parameters():
        assert rel(p.grad, sd[k].grad) < 1e-4, k


def test_submodule_api_matches_oracle():
    """Attention / MLP / PatchEmbed / Block keep the reference's standalone forward signatures."""
    from oracle import mit as omit
    from semseg_amd import segformer as sf
    b, h, w, c = 2, 8, 12, 64
    x = det_input('sub.x', (b, h * w, c), -1, 1)
    for mod, fn in ((sf.Attention(c, 2, 4), lambda sd, t: omit.attention(sd, 'p', t, h, w, 2, 4)),
                    (sf.MLP(c, 4 * c), lambda sd, t: omit.mix_ffn(sd, 'p', t, h, w)),
                    (sf.Block(c, 1, 2, 0.0), lambda sd, t: omit.block(sd, 'p', t, h, w, 1, 2))):
        det_fill_(mod).cuda()
        xd = x.cuda().requires_grad_(True)
        y = mod(xd, h, w)
        y.sum().backward()
        sd = {'p.' + k: v.detach().cpu() for k, v in mod.state_dict().items()}
        xr = x.clone().requires_grad_(True)
        yr = fn(sd, xr)
        yr.sum().backward()
        assert rel(y, yr) < 1e-4 and rel(xd.grad, xr.grad) < 1e-4, type(mod).__name__
    pe = det_fill_(sf.PatchEmbed(64, 128, 3, 2)).cuda()
    img = det_input('sub.img', (2, 64, 10, 14), -1, 1)
    xd = img.cuda().requires_grad_(True)
    y, ho, wo = pe(xd)
    y.sum().backward()
    sd = {'p.' + k: v.detach().cpu() for k, v in pe.state_dict().items()}
    xr = img.clone().requires_grad_(True)
    yr, hr, wr = omit.patch_embed(sd, 'p', xr, 3, 2)
    yr.sum().backward()
    assert (ho, wo) == (hr, wr) and rel(y, yr) < 1e-4 and rel(xd.grad, xr.grad) < 1e-4
