"""The one-kernel Mix-FFN of MiT stages 1-2 (ss_mixffn_fwd / ss_mixffn_bwd; modules/segformer.py:44-64,96-99) against the plain fp32
torch expression of the reference's MLP on the values the kernel sees, and against the three-launch form it replaces
(ss_gemm_nt -> ss_dwconv3_fwd(GELU) -> ss_gemm_nt with residual and DropPath scale)."""
import pytest
import torch
import torch.nn.functional as F

from _parity import rel as _rel

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rel(a, b):
    return _rel(a, b, floor=1e-3)


def _rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


def _case(B, H, W, C, seed=0):
    Ch = 4 * C
    p = dict(x=_rnd(B * H * W, C, seed=seed), w1=_rnd(Ch, C, seed=seed + 1, scale=C ** -0.5), b1=0.2 * _rnd(Ch, seed=seed + 2),
             wd=_rnd(Ch, 1, 3, 3, seed=seed + 3, scale=0.4), bd=0.2 * _rnd(Ch, seed=seed + 4), w2=_rnd(C, Ch, seed=seed + 5, scale=Ch ** -0.5),
             b2=0.2 * _rnd(C, seed=seed + 6), res=_rnd(B * H * W, C, seed=seed + 7), keep=torch.tensor([0.0, 1.25, 1.25, 0.0, 1.25][:B]))
    return p, Ch


def _torch_mlp(p, B, H, W, C, Ch):
    """modules/segformer.py:56-64 in fp32 on bf16-rounded operands, hidden tensors rounded where the bf16 path stores them"""
    r = lambda t: t.to(BF).float()
    x = r(p['x']).requires_grad_(True)
    w1, w2 = r(p['w1']).requires_grad_(True), r(p['w2']).requires_grad_(True)
    b1, bd, b2, wd = (p[k].clone().requires_grad_(True) for k in ('b1', 'bd', 'b2', 'wd'))
    h1 = F.linear(x, w1, b1)
    h1 = h1 + (r(h1.detach()) - h1.detach())
    u = F.conv2d(h1.view(B, H, W, Ch).permute(0, 3, 1, 2), wd, bd, padding=1, groups=Ch)
    a2 = F.gelu(u).permute(0, 2, 3, 1).reshape(B * H * W, Ch)
    a2 = a2 + (r(a2.detach()) - a2.detach())
    y = F.linear(a2, w2, b2)
    out = y * p['keep'].repeat_interleave(H * W)[:, None] + r(p['res'])
    return out, dict(x=x, w1=w1, w2=w2, b1=b1, bd=bd, b2=b2, wd=wd), h1, a2


@pytest.mark.parametrize('B,H,W,C', [(2, 20, 37, 64), (1, 8, 16, 64), (3, 5, 3, 64), (2, 13, 9, 128), (1, 16, 16, 128), (2, 64, 64, 64)])
def test_mixffn_forward(B, H, W, C):
    from semseg_amd import ops
    p, Ch = _case(B, H, W, C)
    d = {k: v.cuda() for k, v in p.items()}
    x, w1, w2, res = (d[k].to(BF) for k in ('x', 'w1', 'w2', 'res'))
    w9c = d['wd'].reshape(Ch, 9).t().contiguous()
    assert ops.mixffn_ok(x, Ch)
    out, h1, a2 = ops.mixffn_fwd(x, w1, d['b1'], w9c, d['bd'], w2, d['b2'], B, H, W, res=res, keep=d['keep'], save_hidden=True)
    out_plain = ops.mixffn_fwd(x, w1, d['b1'], w9c, d['bd'], w2, d['b2'], B, H, W)
    # the three-launch form
    h1_3 = ops.gemm_nt(x, w1, bias=d['b1'])
    a2_3 = ops.dwconv3_fwd(h1_3, w9c, d['bd'], B, H, W, act=1)
    out_3 = ops.gemm_nt(a2_3, w2, bias=d['b2'], res=res, rowscale=d['keep'], rows_per_scale=H * W)
    assert rel(h1, h1_3) < 4e-3 and rel(a2, a2_3) < 8e-3          # the same roundings; accumulation order differs (one bf16 ulp)
    assert rel(out, out_3) < 8e-3
    assert rel(out_plain, ops.gemm_nt(a2_3, w2, bias=d['b2'])) < 8e-3
    ref, _, h1_r, a2_r = _torch_mlp(p, B, H, W, C, Ch)
    assert rel(h1, h1_r) < 4e-3 and rel(a2, a2_r) < 8e-3
    assert rel(out, ref) < 1e-2                                    # bf16 output tolerance of the op tests (tests/test_ops_gpu.py:_tol)
    # dropped samples (keep = 0) return the residual exactly
    hw = H * W
    assert torch.equal(out[:hw], res[:hw])
