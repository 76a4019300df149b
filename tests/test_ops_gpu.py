"""GPU unit tests of the HIP kernels (through the C ABI) against plain fp32 PyTorch-CPU maths of the same op.
Tolerances: _tol / _gtol below; what each site measured is in gpurun_out/parity_errors.json."""
import math
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16]


def _tol(dtype):
    """outputs: fp32 1e-4 (north star); bf16 1e-2 of the tensor's max magnitude (measured <= 4e-3; the north star's 2e-2 is a relative
    L2 bound on whole-network outputs)"""
    return 1e-4 if dtype == torch.float32 else 1e-2


def _gtol(dtype):
    """gradients of a single op: fp32 1e-4, bf16 2e-2 (measured <= 8e-3)"""
    return 1e-4 if dtype == torch.float32 else 2e-2


from _parity import rel as _rel  # noqa: E402


def rel(a, b):
    return _rel(a, b, floor=1e-3)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale)


def dev(t, dtype=None):
    t = t.cuda()
    return t.to(dtype) if dtype is not None else t


def q(t, dtype):
    """the fp32 values the kernel actually sees after storage rounding"""
    return t.to(dtype).float().clone()


@pytest.fixture(scope='module')
def ops():
    from semseg_amd import ops as o
    return o


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('M,N,K', [(300, 64, 64), (257, 320, 160), (128, 6, 512), (1000, 48, 32), (130, 512, 560), (64, 256, 6), (515, 136, 147)])
def test_gemm_nt(ops, dtype, M, N, K):
    a, b, bias, res = rnd(M, K), rnd(N, K, seed=1) / math.sqrt(K), rnd(N, seed=2), rnd(M, N, seed=3)
    scale = torch.tensor([1.0, 0.0, 2.0, 0.5])
    rps = (M + 3) // 4
    ref = (q(a, dtype) @ q(b, dtype).t() + bias) * scale.repeat_interleave(rps)[:M, None] + q(res, dtype)
    out = ops.gemm_nt(dev(a, dtype), dev(b, dtype), bias=dev(bias), res=dev(res, dtype), rowscale=dev(scale), rows_per_scale=rps)
    assert out.dtype == dtype and rel(out, ref) < _tol(dtype)
    out2 = ops.gemm_nt(dev(a, dtype), dev(b, dtype), act=1, out_f32=True)
    assert out2.dtype == torch.float32 and rel(out2, F.gelu(q(a, dtype) @ q(b, dtype).t())) < (1e-4 if dtype == torch.float32 else 1e-3)


@pytest.mark.parametrize('M,N,K', [(1000, 64, 64), (777, 128, 128), (4099, 320, 320), (513, 512, 560), (260, 1280, 320), (300, 48 + 512, 512),
                                   (2050, 256, 2560), (129, 576, 152), (640, 8, 64), (1003, 512, 6), (77, 1032, 8),
                                   (65536 + 77, 256, 256), (33000, 512, 320), (17000, 1024, 576),      # 256 x 256 tile kernel
                                   (131072 + 77, 512, 256), (65536 + 300, 1280, 320), (131072 + 9, 512, 128),   # its persistent form (>= 4 tiles per CU)
                                   (32768 + 77, 320, 1280), (40000, 640, 320), (65536, 320, 256),               # 256 x 320 tiles (N % 320 == 0)
                                   (65536 + 5, 256, 96), (131072, 128, 64), (70000, 256, 32 * 11), (33000, 512, 128),    # long-M shapes of the 128 tile kernel
                                   (262144 + 77, 256, 64), (262144 + 5, 64, 64), (262144 + 33, 128, 64), (262144 + 40, 128, 128),
                                   (262144 + 1, 64, 128), (270000, 256, 128), (262144 + 70, 512, 128)])                         # the streaming kernel (K = 64 / 128, N <= 256, M >= 262144)
def test_gemm_nt_bf16_fast_path(ops, M, N, K):
    """shapes that take the 128 x {64,128} x 64 bf16 kernels: ragged M, K tails (K % 64 != 0), N tails, bias + GELU + DropPath row
    scale + residual in the epilogue, and writes into a channel slice of a wider buffer."""
    dtype = torch.bfloat16
    a, b, bias, res = rnd(M, K), rnd(N, K, seed=1) / math.sqrt(K), rnd(N, seed=2), rnd(M, N, seed=3)
    scale = torch.tensor([1.0, 0.0, 2.0, 0.5])
    rps = (M + 3) // 4
    pre = q(a, dtype) @ q(b, dtype).t() + bias
    ref = pre * scale.repeat_interleave(rps)[:M, None] + q(res, dtype)
    out = ops.gemm_nt(dev(a, dtype), dev(b, dtype), bias=dev(bias), res=dev(res, dtype), rowscale=dev(scale), rows_per_scale=rps)
    assert rel(out, ref) < 1e-2
    wide = torch.full((M, N + 16), 7.0, dtype=dtype, device='cuda')
    ops.gemm_nt(dev(a, dtype), dev(b, dtype), bias=dev(bias), act=1, out=wide[:, 8:8 + N])
    assert rel(wide[:, 8:8 + N], F.gelu(pre)) < 1e-2
    assert float((wide[:, :8] - 7).abs().max()) == 0 and float((wide[:, 8 + N:] - 7).abs().max()) == 0
    big = dev(rnd(M, K + 24), dtype)                          # strided A (a channel slice of a concat buffer)
    out3 = ops.gemm_nt(big[:, 8:8 + K], dev(b, dtype))
    assert rel(out3, big[:, 8:8 + K].float().cpu() @ q(b, dtype).t()) < 1e-2


@pytest.mark.parametrize('M,N,K', [(70001, 8, 8), (5000, 8, 32), (65536, 32, 8), (4100, 16, 16), (9000, 32, 32), (4096, 16, 8), (6000, 8, 16), (5000, 16, 32)])
@pytest.mark.parametrize('dtype', DTYPES)
def test_gemm_nt_both_sides_narrow(ops, dtype, M, N, K):
    """K, N in {8, 16, 32} (the 8-channel maps of the RefinerHead, fp32 there): the thread-per-row kernel, dense and into / out of channel slices."""
    a, b, bias = rnd(M, K), rnd(N, K, seed=1) / math.sqrt(K), rnd(N, seed=2)
    ref = q(a, dtype) @ q(b, dtype).t() + bias
    tol = 1e-2 if dtype == torch.bfloat16 else 1e-5
    assert rel(ops.gemm_nt(dev(a, dtype), dev(b, dtype), bias=dev(bias)), ref) < tol
    assert rel(ops.gemm_nt(dev(a, dtype), dev(b, dtype)), ref - bias) < tol
    wide = torch.full((M, N + 16), 7.0, dtype=dtype, device='cuda')
    big = dev(rnd(M, K + 24, seed=4), dtype)
    ops.gemm_nt(big[:, 8:8 + K], dev(b, dtype), bias=dev(bias), out=wide[:, 8:8 + N])
    assert rel(wide[:, 8:8 + N], big[:, 8:8 + K].float().cpu() @ q(b, dtype).t() + bias) < tol
    assert float((wide[:, :8] - 7).abs().max()) == 0 and float((wide[:, 8 + N:] - 7).abs().max()) == 0


@pytest.mark.parametrize('R,N', [(5000, 8), (70001, 32), (4097, 16), (300000, 8)])
@pytest.mark.parametrize('dtype', DTYPES)
def test_gemm_tn_both_sides_narrow(ops, dtype, R, N):
    """[R, 8]^T . [R, N], N <= 32: a lane per row instead of a wave per row; strided operands; bit-reproducible."""
    a, b = rnd(R, 8), rnd(R, N, seed=1)
    ref = q(a, dtype).t() @ q(b, dtype)
    tol = 2e-3 if dtype == torch.bfloat16 else 1e-4
    out = ops.gemm_tn(dev(a, dtype), dev(b, dtype))
    assert rel(out, ref) < tol
    assert torch.equal(out, ops.gemm_tn(dev(a, dtype), dev(b, dtype)))
    big_a, big_b = dev(rnd(R, 24, seed=3), dtype), dev(rnd(R, N + 8, seed=5), dtype)
    out2 = ops.gemm_tn(big_a[:, 8:16], big_b[:, :N])
    assert rel(out2, big_a[:, 8:16].float().cpu().t() @ big_b[:, :N].float().cpu()) < tol
    out3, cs = ops.gemm_tn(dev(a, dtype), dev(b, dtype), colsum=True)
    assert torch.equal(out3, out) and rel(cs, q(a, dtype).sum(0)) < tol


@pytest.mark.parametrize('R,M,N', [(5000, 64, 64), (4097, 320, 1280), (9000, 512, 48 + 512), (1024, 128, 2048), (70, 8, 16), (3000, 1152, 320),
                                   (70001, 6, 512), (333, 3, 1032),
                                   (8192, 512, 256), (4096, 384, 512), (12288, 256, 328), (16384, 520, 768),     # 256-tile kernel, ragged tiles
                                   (8192, 1280, 320), (8192, 320, 1280), (4096, 328, 640), (12288, 640, 264),    # 256 x 320 / 320 x 256 tiles
                                   (8192, 512, 128), (16384, 128, 512), (65536, 256, 64), (32768, 64, 256), (8192, 128, 128),
                                   (8192, 448, 104), (4096, 56, 208)])                                           # one tile = the whole output
def test_gemm_tn_bf16_fast_path(ops, R, M, N):
    dtype = torch.bfloat16
    a, b = rnd(R, M), rnd(R, N, seed=1)
    ref = q(a, dtype).t() @ q(b, dtype)
    out = ops.gemm_tn(dev(a, dtype), dev(b, dtype))
    assert rel(out, ref) < 2e-3
    big_a, big_b = dev(rnd(R, M + 16), dtype), dev(rnd(R, N + 8, seed=5), dtype)     # strided operands
    out2 = ops.gemm_tn(big_a[:, 8:8 + M], big_b[:, :N])
    assert rel(out2, big_a[:, 8:8 + M].float().cpu().t() @ big_b[:, :N].float().cpu()) < 2e-3
    assert torch.equal(out, ops.gemm_tn(dev(a, dtype), dev(b, dtype)))


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('R,M,N', [(5000, 64, 64), (70001, 320, 136), (9000, 512, 48 + 512), (300, 8, 16), (4097, 6, 512), (100000, 1280, 320),
                                   (8192, 512, 512), (4096, 328, 256), (65536, 1280, 320), (8192, 320, 1280), (8192, 640, 1000),
                                   (65536, 256, 64), (16384, 512, 128), (8192, 104, 448), (8192, 64, 256), (4096, 128, 128)])
def test_gemm_tn_with_bias_gradient(ops, dtype, R, M, N):
    """ss_gemm_tn_colsum: dW = dY^T X and db = column sums of dY from one pass (fused in the bf16 MFMA kernel, a second launch on
    the other paths); must agree with the two separate entry points."""
    a, b = rnd(R, M), rnd(R, N, seed=1)
    out, cs = ops.gemm_tn(dev(a, dtype), dev(b, dtype), colsum=True)
    ref = q(a, dtype).t() @ q(b, dtype)
    assert rel(out, ref) < (1e-2 if dtype == torch.bfloat16 else 1e-4)
    assert float((out - ops.gemm_tn(dev(a, dtype), dev(b, dtype))).abs().max()) == 0
    refcs = q(a, dtype).double().sum(0).float()
    assert float((cs.cpu() - refcs).abs().max()) <= 1e-4 * float(q(a, dtype).abs().double().sum(0).max())


@pytest.mark.parametrize('R,M,N,rpk', [(4096, 64, 256, 1024), (8192, 320, 1280, 1024), (16384, 512, 2048, 1024), (65536, 128, 128, 4096),
                                       (1024, 64, 64, 64), (12288, 512, 512, 4096), (8192, 1280, 320, 1024),
                                       (131072, 512, 128, 8192), (65536, 256, 64, 1024), (16384, 128, 512, 256), (16384, 512, 512, 256)])
def test_gemm_tn_droppath_keep(ops, R, M, N, rpk):
    """ss_gemm_tn_colsum_keep: weight / bias gradient under DropPath (modules/helpers.py:9-24) from the unscaled gradient; equals the plain
    entry point on the row-scaled gradient up to the rounding of that scaled copy, and is exactly zero when every sample is dropped."""
    dtype = torch.bfloat16
    a, b = rnd(R, M), rnd(R, N, seed=1)
    nb = R // rpk
    keep = (torch.rand(nb, generator=torch.Generator().manual_seed(R)) > 0.35).float() / 0.65
    keep[0] = 0.0
    if nb > 1:
        keep[-1] = 1 / 0.65
    ad, bd = dev(a, dtype), dev(b, dtype)
    assert ops.gemm_tn_keep_ok(ad, bd, rpk)
    dW, db = ops.gemm_tn_keep(ad, bd, dev(keep), rpk)
    k = keep.repeat_interleave(rpk)[:, None]
    ref = (q(a, dtype) * k).t() @ q(b, dtype)
    assert rel(dW, ref) < 2e-3
    assert rel(db, (q(a, dtype) * k).sum(0)) < 2e-3
    dW0, db0 = ops.gemm_tn_keep(ad, bd, torch.zeros(nb, device='cuda'), rpk)
    assert float(dW0.abs().max()) == 0 and float(db0.abs().max()) == 0
    assert not ops.gemm_tn_keep_ok(ad, bd, 96) and not ops.gemm_tn_keep_ok(ad.float(), bd.float(), rpk)


@pytest.mark.parametrize('M,N,K', [(16, 512, 512), (2, 512, 512), (1, 64, 64), (7, 200, 1032), (16, 6, 512)])
def test_gemm_nt_few_rows(ops, M, N, K):
    """M <= 16 rows in bf16 (the per-image vectors of the image-pool branch): the wave-per-column kernel, both output dtypes"""
    dtype = torch.bfloat16
    a, b, bias = rnd(M, K), rnd(N, K, seed=1) / math.sqrt(K), rnd(N, seed=2)
    ref = q(a, dtype) @ q(b, dtype).t() + bias
    out = ops.gemm_nt(dev(a, dtype), dev(b, dtype), bias=dev(bias))
    assert out.dtype == dtype and rel(out, ref) < 1e-2
    out32 = ops.gemm_nt(dev(a, dtype), dev(b, dtype), bias=dev(bias), out_f32=True)
    assert out32.dtype == torch.float32 and rel(out32, ref) < 1e-5
    wide = torch.full((M, N + 16), 7.0, dtype=torch.float32, device='cuda')
    ops.gemm_nt(dev(a, dtype), dev(b, dtype), out=wide[:, 8:8 + N], out_f32=True)
    assert rel(wide[:, 8:8 + N], ref - bias) < 1e-5 and float((wide[:, :8] - 7).abs().max()) == 0 and float((wide[:, 8 + N:] - 7).abs().max()) == 0


@pytest.mark.parametrize('dtype', DTYPES)
def test_gemm_nt_strided(ops, dtype):
    big_a, b = rnd(200, 96), rnd(40, 32, seed=1)
    a = dev(big_a, dtype)[:, 32:64]
    outbuf = torch.zeros(200, 128, dtype=dtype, device='cuda')
    ops.gemm_nt(a, dev(b, dtype), out=outbuf[:, 40:80])
    ref = q(big_a[:, 32:64], dtype) @ q(b, dtype).t()
    assert rel(outbuf[:, 40:80], ref) < _tol(dtype)
    assert float(outbuf[:, :40].abs().max()) == 0 and float(outbuf[:, 80:].abs().max()) == 0


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('R,M,N', [(1000, 64, 64), (4097, 320, 160), (513, 6, 512), (300, 48, 32), (2048, 512, 136), (77, 8, 8)])
def test_gemm_tn(ops, dtype, R, M, N):
    a, b = rnd(R, M), rnd(R, N, seed=1)
    ref = q(a, dtype).t() @ q(b, dtype)
    out = ops.gemm_tn(dev(a, dtype), dev(b, dtype))
    assert rel(out, ref) < (1e-4 if dtype == torch.float32 else 2e-3)
    out2 = ops.gemm_tn(dev(a, dtype), dev(b, dtype), out=out.clone(), accumulate=True)
    assert rel(out2, 2 * ref) < (1e-4 if dtype == torch.float32 else 2e-3)
    # bit-exact reproducibility (fixed-order slab reduction, no atomics)
    assert torch.equal(out, ops.gemm_tn(dev(a, dtype), dev(b, dtype)))


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('rows,C', [(1000, 32), (777, 64), (300, 160), (129, 320), (64, 512), (50, 768)])
def test_layernorm(ops, dtype, rows, C):
    x, g, b, dy = rnd(rows, C, scale=2.0) + 0.5, 1 + 0.1 * rnd(C, seed=1), 0.1 * rnd(C, seed=2), rnd(rows, C, seed=3)
    xr = q(x, dtype).requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gr, br, 1e-5)
    ref.backward(q(dy, dtype))
    y, mean, rstd = ops.layernorm_fwd(dev(x, dtype), dev(g), dev(b), 1e-5)
    assert rel(y, ref) < _tol(dtype)
    dx, dgb = ops.layernorm_bwd(dev(dy, dtype), dev(x, dtype), mean, rstd, dev(g))
    assert rel(dx, xr.grad) < _tol(dtype)
    assert rel(dgb[0], gr.grad) < 1e-4 and rel(dgb[1], br.grad) < 1e-4
    res = rnd(rows, C, seed=4)                               # gradient of the residual path, added in the same pass
    dx2, dgb2 = ops.layernorm_bwd(dev(dy, dtype), dev(x, dtype), mean, rstd, dev(g), res=dev(res, dtype))
    assert rel(dx2, xr.grad + q(res, dtype)) < _tol(dtype)
    assert float((dgb2 - dgb).abs().max()) == 0


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('rows,C,act', [(2000, 64, 1), (333, 128, 0), (4, 256, 1), (70, 48, 1)])
def test_batchnorm(ops, dtype, rows, C, act):
    B = 2 if rows % 2 == 0 else 1
    z, g, b, dy = rnd(rows, C, scale=1.5) + 0.3, 1 + 0.1 * rnd(C, seed=1), 0.1 * rnd(C, seed=2), rnd(rows, C, seed=3)
    res = rnd(rows, C, seed=4)
    mask = (torch.rand(B, C, generator=torch.Generator().manual_seed(1)) > 0.3).float() / 0.7
    rm, rv = 0.1 * rnd(C, seed=5), 1 + 0.1 * rnd(C, seed=6).abs()
    zr = q(z, dtype).requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    u = F.batch_norm(zr.t()[None], rm_ref, rv_ref, gr, br, True, 0.1, 1e-5)[0].t()
    ref = (F.mish(u) if act else u) * mask.repeat_interleave(rows // B, 0) + q(res, dtype)
    ref.backward(q(dy, dtype))
    rm_d, rv_d, nbt = dev(rm), dev(rv), torch.zeros((), dtype=torch.int64, device='cuda')
    stats = ops.bn_stats(dev(z, dtype), 1e-5, 0.1, rm_d, rv_d, nbt)
    assert rel(rm_d, rm_ref) < 1e-4 and rel(rv_d, rv_ref) < 1e-4 and int(nbt) == 1
    y = ops.bn_apply(dev(z, dtype), stats, dev(g), dev(b), act, res=dev(res, dtype), mask=dev(mask), rows_per_image=rows // B)
    assert rel(y, ref) < _tol(dtype)
    dz, dgb = ops.bn_bwd(dev(dy, dtype), dev(z, dtype), stats, dev(g), dev(b), act, mask=dev(mask), rows_per_image=rows // B)
    assert rel(dz, zr.grad) < _gtol(dtype)
    assert rel(dgb[0], gr.grad) < 1e-4 and rel(dgb[1], br.grad) < 1e-4
    # eval mode
    zr2 = q(z, dtype).requires_grad_(True)
    ref2 = F.batch_norm(zr2.t()[None], rm, rv, g, b, False, 0.1, 1e-5)[0].t()
    ref2.backward(q(dy, dtype))
    st2 = ops.bn_eval_stats(dev(rm), dev(rv))
    assert rel(ops.bn_apply(dev(z, dtype), st2, dev(g), dev(b), 0), ref2) < _tol(dtype)
    dz2, _ = ops.bn_bwd(dev(dy, dtype), dev(z, dtype), st2, dev(g), dev(b), 0, training=False)
    assert rel(dz2, zr2.grad) < _tol(dtype)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('M,N,K,with_res', [(1000, 64, 64, False), (4099, 512, 560, False), (33000, 512, 256, True), (70000, 256, 512, False),
                                            (131072 + 9, 512, 128, True), (300, 48, 32, False), (2, 512, 512, False)])
def test_gemm_with_batchnorm_statistics(ops, dtype, M, N, K, with_res):
    """ss_gemm_nt_bn_stats (conv -> BatchNorm2d of modules/lawin.py:49-54,97-105): the product is the one ss_gemm_nt gives, bit for bit, and
    the statistics / running buffers agree with ss_bn_stats of that product (tile column sums from the GEMM's copy-out on the bf16 tile
    kernels -- 128-row, 256-row and the shapes that would otherwise take the persistent kernel -- the two calls elsewhere)."""
    a, b = rnd(M, K), rnd(N, K, seed=1) / math.sqrt(K)
    res = dev(rnd(M, N, seed=3) + 0.5, dtype) if with_res else None
    ad, bd = dev(a, dtype), dev(b, dtype)
    z0 = ops.gemm_nt(ad, bd, res=res)
    rm0, rv0, n0 = torch.zeros(N, device='cuda'), torch.ones(N, device='cuda'), torch.zeros((), dtype=torch.int64, device='cuda')
    st0 = ops.bn_stats(z0, 1e-5, 0.1, rm0, rv0, n0)
    rm1, rv1, n1 = torch.zeros(N, device='cuda'), torch.ones(N, device='cuda'), torch.zeros((), dtype=torch.int64, device='cuda')
    z1, st1 = ops.gemm_nt_bn_stats(ad, bd, res, 1e-5, 0.1, rm1, rv1, n1)
    assert torch.equal(z0, z1) and int(n1) == 1
    assert rel(st1[0], st0[0]) < 1e-5 and rel(st1[1], st0[1]) < 1e-4 and rel(rm1, rm0) < 1e-5 and rel(rv1, rv0) < 1e-4
    zf = z0.float().cpu().double()
    assert rel(st1[0], zf.mean(0).float()) < 1e-5 and rel(st1[1], (zf.var(0, unbiased=False) + 1e-5).rsqrt().float()) < 1e-4


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('B,hi,wi,ho,wo,C,act', [(2, 8, 8, 16, 16, 64, 1), (1, 5, 7, 10, 14, 128, 0), (3, 6, 4, 13, 9, 48, 1), (2, 16, 16, 32, 32, 512, 1),
                                                 (1, 1, 3, 2, 6, 16, 0), (2, 50, 3, 100, 6, 1032, 1),
                                                 (1, 64, 2, 128, 4, 64, 1)])     # thin map: more x2 walker blocks than column-geometry blocks
def test_batchnorm_of_bilinear_resize(ops, dtype, B, hi, wi, ho, wo, C, act):
    """ss_bn_stats_up / ss_bn_apply_up / ss_bn_bwd_up (LawinHead `cat`: BatchNorm + Mish of a bilinear resize, modules/lawin.py:213-222,
    101-105) against F.interpolate + F.batch_norm (+ F.mish) in fp32, and bit for bit against the unfused pair ss_bilinear_fwd +
    ss_bn_stats / ss_bn_apply / ss_bn_bwd that they replace."""
    x = rnd(B * hi * wi, C, scale=1.5) + 0.3
    g, b, dy = 1 + 0.1 * rnd(C, seed=1), 0.1 * rnd(C, seed=2), rnd(B * ho * wo, C, seed=3)
    rm, rv = 0.1 * rnd(C, seed=5), 1 + 0.1 * rnd(C, seed=6).abs()
    xd, dyd = dev(x, dtype), dev(dy, dtype)
    # the unfused pair
    z = ops.bilinear_fwd(xd, B, hi, wi, ho, wo)
    rm1, rv1, n1 = dev(rm), dev(rv), torch.zeros((), dtype=torch.int64, device='cuda')
    st1 = ops.bn_stats(z, 1e-5, 0.1, rm1, rv1, n1)
    y1 = ops.bn_apply(z, st1, dev(g), dev(b), act)
    dz1, dgb1 = ops.bn_bwd(dyd, z, st1, dev(g), dev(b), act)
    # fused: writes into a channel slice of a wider buffer, as the head does
    rm2, rv2, n2 = dev(rm), dev(rv), torch.zeros((), dtype=torch.int64, device='cuda')
    st2 = ops.bn_stats_up(xd, B, hi, wi, ho, wo, 1e-5, 0.1, rm2, rv2, n2)
    # statistics: the same values summed in another order (the x2 case walks image columns, not matrix rows)
    assert rel(st2, st1) < 1e-5 and rel(rm2, rm1) < 1e-5 and rel(rv2, rv1) < 1e-5 and int(n2) == 1
    wide = torch.full((B * ho * wo, C + 16), 7.0, dtype=dtype, device='cuda')
    ops.bn_apply_up(xd, B, hi, wi, ho, wo, st1, dev(g), dev(b), act, out=wide[:, 8:8 + C])
    assert torch.equal(y1, wide[:, 8:8 + C])                  # same statistics in -> the same bits out
    assert float((wide[:, :8] - 7).abs().max()) == 0 and float((wide[:, 8 + C:] - 7).abs().max()) == 0
    dz2, dgb2 = ops.bn_bwd_up(dyd, xd, B, hi, wi, ho, wo, st1, dev(g), dev(b), act)
    assert rel(dgb2, dgb1) < 1e-5 and rel(dz2, dz1) < (1e-5 if dtype == torch.float32 else 4e-3)
    # torch fp32 on the values the kernel sees; the resized map is rounded to the storage dtype as the reference's autocast output is
    xr = q(x, dtype).requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    zi = F.interpolate(xr.view(B, hi, wi, C).permute(0, 3, 1, 2), size=(ho, wo), mode='bilinear', align_corners=False)
    zq = zi + (zi.detach().to(dtype).float() - zi.detach())
    u = F.batch_norm(zq, rm.clone(), rv.clone(), gr, br, True, 0.1, 1e-5)
    ref = (F.mish(u) if act else u).permute(0, 2, 3, 1).reshape(B * ho * wo, C)
    ref.backward(q(dy, dtype))
    assert rel(wide[:, 8:8 + C], ref) < _tol(dtype)
    assert rel(dgb2[0], gr.grad) < (1e-4 if dtype == torch.float32 else 2e-3) and rel(dgb2[1], br.grad) < (1e-4 if dtype == torch.float32 else 2e-3)
    dx = ops.bilinear_bwd(dz2, B, hi, wi, ho, wo)
    assert rel(dx, xr.grad) < _gtol(dtype)
    # second pass with the transpose of the resize folded in (x2), the unfused pair otherwise
    dx2, dgb5 = ops.bn_bwd_up_to_input(dyd, xd, B, hi, wi, ho, wo, st1, dev(g), dev(b), act)
    assert rel(dx2, xr.grad) < _gtol(dtype) and rel(dx2, dx) < (1e-5 if dtype == torch.float32 else 8e-3) and rel(dgb5, dgb2) < 1e-6
    # the same with dy given up as scratch: pass 1 leaves dy * act'(.) in it (rounded to the storage dtype), pass 2 reads that
    scratch = torch.full((B * ho * wo, C + 16), 7.0, dtype=dtype, device='cuda')
    scratch[:, 8:8 + C] = dyd
    dx6, dgb6 = ops.bn_bwd_up_to_input(scratch[:, 8:8 + C], xd, B, hi, wi, ho, wo, st1, dev(g), dev(b), act, overwrite_dy=True)
    assert rel(dx6, xr.grad) < _gtol(dtype) and rel(dx6, dx2) < (1e-6 if dtype == torch.float32 else 8e-3)
    assert rel(dgb6, dgb5) < (1e-6 if dtype == torch.float32 else 2e-3)
    assert float((scratch[:, :8] - 7).abs().max()) == 0 and float((scratch[:, 8 + C:] - 7).abs().max()) == 0
    if not act or (ho, wo) != (2 * hi, 2 * wi):
        assert torch.equal(scratch[:, 8:8 + C], dyd)          # nothing to keep: dy is left alone
    # eval mode
    st3 = ops.bn_eval_stats(dev(rm), dev(rv))
    y3 = ops.bn_apply_up(xd, B, hi, wi, ho, wo, st3, dev(g), dev(b), act)
    assert torch.equal(y3, ops.bn_apply(z, st3, dev(g), dev(b), act))
    dz3, dgb3 = ops.bn_bwd_up(dyd, xd, B, hi, wi, ho, wo, st3, dev(g), dev(b), act, training=False)
    dz4, dgb4 = ops.bn_bwd(dyd, z, st3, dev(g), dev(b), act, training=False)
    assert torch.equal(dz3, dz4) and rel(dgb3, dgb4) < 1e-5   # eval mode: dz does not depend on the sums


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('B,hi,wi,C,act', [(2, 8, 8, 64, 1), (3, 5, 7, 128, 0), (2, 16, 12, 512, 1)])
def test_batchnorm_of_resize_with_per_image_constant(ops, dtype, B, hi, wi, C, act):
    """bvec of ss_bn_*_up: BatchNorm (+ Mish) of resize(x) + v[image] (the image-pool branch of LawinHead's `cat` conv as one vector per
    image, modules/lawin.py:205-222) against torch fp32; the gradient of v is the per-image column sum of the returned dx."""
    ho, wo = 2 * hi, 2 * wi
    x, v = rnd(B * hi * wi, C, scale=1.5) + 0.3, rnd(B, C, seed=11)
    g, b, dy = 1 + 0.1 * rnd(C, seed=1), 0.1 * rnd(C, seed=2), rnd(B * ho * wo, C, seed=3)
    xr, vr = q(x, dtype).requires_grad_(True), v.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    zi = F.interpolate(xr.view(B, hi, wi, C).permute(0, 3, 1, 2), size=(ho, wo), mode='bilinear', align_corners=False)
    zq = zi + (zi.detach().to(dtype).float() - zi.detach()) + vr[:, :, None, None]
    rm, rv = torch.zeros(C), torch.ones(C)
    u = F.batch_norm(zq, rm, rv, gr, br, True, 0.1, 1e-5)
    ref = (F.mish(u) if act else u).permute(0, 2, 3, 1).reshape(B * ho * wo, C)
    ref.backward(q(dy, dtype))
    xd, dyd, vd = dev(x, dtype), dev(dy, dtype), dev(v)
    rm2, rv2 = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    st = ops.bn_stats_up(xd, B, hi, wi, ho, wo, 1e-5, 0.1, rm2, rv2, None, bvec=vd)
    assert rel(rm2, rm) < 1e-4 and rel(rv2, rv) < 1e-4
    y = ops.bn_apply_up(xd, B, hi, wi, ho, wo, st, dev(g), dev(b), act, bvec=vd)
    assert rel(y, ref) < _tol(dtype)
    dz, dgb = ops.bn_bwd_up(dyd, xd, B, hi, wi, ho, wo, st, dev(g), dev(b), act, bvec=vd)
    dx, dgb2 = ops.bn_bwd_up_to_input(dyd, xd, B, hi, wi, ho, wo, st, dev(g), dev(b), act, bvec=vd)
    wtol = 1e-4 if dtype == torch.float32 else 2e-3
    assert rel(dgb[0], gr.grad) < wtol and rel(dgb[1], br.grad) < wtol and rel(dgb2, dgb) < 1e-6
    assert rel(dx, xr.grad) < _gtol(dtype) and rel(ops.bilinear_bwd(dz, B, hi, wi, ho, wo), xr.grad) < _gtol(dtype)
    assert rel(ops.colsum(dx, groups=B), vr.grad) < (1e-4 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('rows,C,nc,masked', [(2000, 64, 6, True), (333, 128, 8, False), (4100, 512, 6, True), (70, 8, 2, False), (1026, 256, 3, True)])
def test_batchnorm_mish_classifier(ops, dtype, rows, C, nc, masked):
    """ss_bn_apply_classify / ss_bn_bwd_classify (LawinHead tail, modules/lawin.py:101-105,227-228: BatchNorm2d + Mish + Dropout2d +
    1x1 classifier) against torch fp32 and against the unfused ss_bn_apply + ss_gemm_nt / ss_gemm_tn + ss_bn_bwd sequence."""
    B = 2 if rows % 2 == 0 else 1
    z, g, b = rnd(rows, C, scale=1.5) + 0.3, 1 + 0.1 * rnd(C, seed=1), 0.1 * rnd(C, seed=2)
    wp, bp, dseg = rnd(nc, C, seed=7) / math.sqrt(C), 0.1 * rnd(nc, seed=8), rnd(rows, nc, seed=9)
    mask = (torch.rand(B, C, generator=torch.Generator().manual_seed(1)) > 0.3).float() / 0.7 if masked else None
    rpi = rows // B if masked else 0
    zr = q(z, dtype).requires_grad_(True)
    gr, br, wr = g.clone().requires_grad_(True), b.clone().requires_grad_(True), wp.clone().requires_grad_(True)
    u = F.mish(F.batch_norm(zr.t()[None], None, None, gr, br, True, 0.1, 1e-5)[0].t())
    if masked:
        u = u * mask.repeat_interleave(rows // B, 0)
    ref = u @ wr.t() + bp
    ref.backward(q(dseg, dtype))
    zd, md = dev(z, dtype), (dev(mask) if masked else None)
    stats = ops.bn_stats(zd)
    seg = ops.bn_apply_classify(zd, stats, dev(g), dev(b), dev(wp), dev(bp), mask=md, rows_per_image=rpi)
    assert seg.dtype == dtype and rel(seg, ref) < _tol(dtype)
    dz, dgb, dwp = ops.bn_bwd_classify(dev(dseg, dtype), zd, stats, dev(g), dev(b), dev(wp), mask=md, rows_per_image=rpi)
    assert rel(dz, zr.grad) < _gtol(dtype)
    wtol = 1e-4 if dtype == torch.float32 else 2e-3          # bf16: the logit gradients are bf16, the sums fp32
    assert rel(dgb[0], gr.grad) < wtol and rel(dgb[1], br.grad) < wtol and rel(dwp, wr.grad) < wtol
    # the unfused sequence on the same inputs (it rounds the activation and its gradient to the storage dtype)
    fused = ops.bn_apply(zd, stats, dev(g), dev(b), 1, mask=md, rows_per_image=rpi)
    seg2 = ops.gemm_nt(fused, dev(wp, dtype), bias=dev(bp))
    dfused = ops.gemm_nt(dev(dseg, dtype), dev(wp.t().contiguous(), dtype))
    dz2, dgb2 = ops.bn_bwd(dfused, zd, stats, dev(g), dev(b), 1, mask=md, rows_per_image=rpi)
    assert rel(seg, seg2) < _tol(dtype) and rel(dz, dz2) < _gtol(dtype) and rel(dgb, dgb2) < (1e-4 if dtype == torch.float32 else 1e-2)
    # eval mode: running statistics, dz without the batch terms
    rm, rv = 0.1 * rnd(C, seed=5), 1 + 0.1 * rnd(C, seed=6).abs()
    zr2 = q(z, dtype).requires_grad_(True)
    u2 = F.mish(F.batch_norm(zr2.t()[None], rm, rv, g, b, False, 0.1, 1e-5)[0].t())
    ref2 = u2 @ wp.t() + bp
    ref2.backward(q(dseg, dtype))
    st2 = ops.bn_eval_stats(dev(rm), dev(rv))
    assert rel(ops.bn_apply_classify(zd, st2, dev(g), dev(b), dev(wp), dev(bp)), ref2) < _tol(dtype)
    dz3, _, _ = ops.bn_bwd_classify(dev(dseg, dtype), zd, st2, dev(g), dev(b), dev(wp), training=False)
    assert rel(dz3, zr2.grad) < _gtol(dtype)


@pytest.mark.parametrize('dtype', DTYPES)
def test_colsum(ops, dtype):
    x = rnd(3 * 500, 64)
    out = ops.colsum(dev(x, dtype), groups=3, scale=0.5)
    assert rel(out, 0.5 * q(x, dtype).view(3, 500, 64).sum(1)) < 1e-4
    wide = rnd(100, 96)
    out = ops.colsum(dev(wide, dtype)[:, 32:64])
    assert rel(out[0], q(wide, dtype)[:, 32:64].sum(0)) < 1e-4


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('B,H,W,C', [(2, 9, 7, 128), (1, 16, 16, 256), (2, 4, 4, 1024), (1, 5, 33, 8), (1, 40, 70, 64), (2, 19, 31, 24), (1, 64, 64, 256)])
def test_dwconv3(ops, dtype, B, H, W, C):
    x, w, b, dy = rnd(B, H, W, C), rnd(C, 1, 3, 3, seed=1) / 3, 0.1 * rnd(C, seed=2), rnd(B, H, W, C, seed=3)
    xr = q(x, dtype).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.gelu(F.conv2d(xr, wr, br, padding=1, groups=C))
    ref.backward(q(dy, dtype).permute(0, 3, 1, 2))
    w9c = ops.transpose_cast(dev(w.view(C, 9)), torch.float32)
    w9c_flip = ops.transpose_cast(dev(w.view(C, 9)), torch.float32, reverse_cols=True)
    xd, dyd = dev(x, dtype).view(-1, C), dev(dy, dtype).view(-1, C)
    y = ops.dwconv3_fwd(xd, w9c, dev(b), B, H, W, act=1)
    assert rel(y.view(B, H, W, C), ref.permute(0, 2, 3, 1)) < _tol(dtype)
    g, dwdb = ops.dwconv3_bwd(dyd, xd, w9c, dev(b), B, H, W, act=1)
    dx = ops.dwconv3_fwd(g, w9c_flip, None, B, H, W, act=0)
    assert rel(dx.view(B, H, W, C), xr.grad.permute(0, 2, 3, 1)) < _gtol(dtype)
    assert rel(dwdb[:9].t().reshape(C, 1, 3, 3), wr.grad) < _gtol(dtype)
    assert rel(dwdb[9], br.grad) < _gtol(dtype)
    # the one-pass backward: same gradients without the stored g (strips with halo rows, column tiles with halo columns)
    dx2, dwdb2 = ops.dwconv3_bwd_fused(dyd, xd, w9c, dev(b), B, H, W, act=1)
    assert rel(dx2.view(B, H, W, C), xr.grad.permute(0, 2, 3, 1)) < _gtol(dtype)
    assert rel(dwdb2[:9].t().reshape(C, 1, 3, 3), wr.grad) < _gtol(dtype)
    assert rel(dwdb2[9], br.grad) < _gtol(dtype)
    if dtype == torch.bfloat16:     # against the two-pass form: the summation order differs, and the one-pass kernel interpolates GELU' in a
        # table (2.5e-5 absolute), which moves an occasional g across a bf16 rounding boundary (32 terms per sum at the smallest shape)
        assert rel(dx2, dx) < 1e-2 and rel(dwdb2, dwdb) < 3e-3


@pytest.mark.parametrize('keep', [False, True])
@pytest.mark.parametrize('B,H,W,C,K', [(2, 40, 70, 256, 64), (3, 19, 33, 512, 128), (1, 64, 64, 64, 64), (2, 7, 5, 128, 128)])
def test_dwconv3_bwd_with_fc2_data_gradient(ops, monkeypatch, B, H, W, C, K, keep):
    """ss_dwconv3_bwd_fc2 == data-gradient GEMM of fc2 (rows scaled by DropPath) followed by the fused depthwise backward: the product
    is rounded to bf16 where the GEMM stored it, so the two differ only by the summation order inside a 64 / 128-term dot product."""
    T = torch.bfloat16
    x, dout = dev(rnd(B * H * W, C), T), dev(rnd(B * H * W, K, seed=1), T)
    w2t = dev(rnd(C, K, seed=2) / K ** 0.5, T)
    w9c, b = dev(rnd(9, C, seed=3) / 3), dev(0.1 * rnd(C, seed=4))
    kp = dev(torch.tensor([0.0, 1.25, 1.25][:B] if B > 1 else [1.25])) if keep else None
    monkeypatch.setenv('SS_DW_FC2', '2')             # K = 128 is off by default (slower than GEMM + depthwise backward there)
    assert ops.dwconv3_bwd_fc2_ok(dout, x, B, H, W)
    dx, dwdb = ops.dwconv3_bwd_fc2(dout, w2t, kp, x, w9c, b, B, H, W)
    da2 = ops.gemm_nt(dout, w2t, rowscale=kp, rows_per_scale=H * W if keep else 0)
    for lane in ('1', '0'):
        monkeypatch.setenv('SS_DW_LANE', lane)
        dx2, dwdb2 = ops.dwconv3_bwd_fused(da2, x, w9c, b, B, H, W, act=1)
        assert rel(dx, dx2) < 4e-3 and rel(dwdb, dwdb2) < 2e-3
    # and against autograd in fp32 on the bf16-rounded operands
    xr = x.float().cpu().view(B, H, W, C).permute(0, 3, 1, 2).requires_grad_(True)
    wr = w9c.cpu().t().reshape(C, 1, 3, 3).clone().requires_grad_(True)
    br = b.cpu().clone().requires_grad_(True)
    dref = dout.float().cpu() @ w2t.float().cpu().t()
    if keep:
        dref = dref * kp.cpu().repeat_interleave(H * W)[:, None]
    F.gelu(F.conv2d(xr, wr, br, padding=1, groups=C)).backward(dref.view(B, H, W, C).permute(0, 3, 1, 2))
    assert rel(dx.view(B, H, W, C), xr.grad.permute(0, 2, 3, 1)) < _gtol(T)
    assert rel(dwdb[:9].t().reshape(C, 1, 3, 3), wr.grad) < _gtol(T)
    assert rel(dwdb[9], br.grad) < _gtol(T)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('B,N,Nk,heads,D', [(2, 256, 4, 1, 32), (1, 1000, 100, 2, 64), (2, 64, 64, 5, 32), (1, 96, 256, 8, 64), (1, 33, 1, 1, 64)])
def test_mit_attention(ops, dtype, B, N, Nk, heads, D):
    C = heads * D
    qq, kv, do = rnd(B * N, C), rnd(B * Nk, 2 * C, seed=1), rnd(B * N, C, seed=2)
    qr, kvr = q(qq, dtype).requires_grad_(True), q(kv, dtype).requires_grad_(True)
    qh = qr.view(B, N, heads, D).transpose(1, 2)
    k, v = kvr.view(B, Nk, 2, heads, D).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(qh @ k.transpose(-2, -1) * D ** -0.5, -1) @ v).transpose(1, 2).reshape(B * N, C)
    ref.backward(q(do, dtype))
    o, lse = ops.mit_attn_fwd(dev(qq, dtype), dev(kv, dtype), B, N, Nk, heads)
    assert rel(o, ref) < _tol(dtype)
    dq, dkv = ops.mit_attn_bwd(dev(qq, dtype), dev(kv, dtype), o, dev(do, dtype), lse, B, N, Nk, heads)
    # Nk == 1 makes dq analytically zero: compare against the gradient scale of the op instead of max|ref| = 0
    scale_q = max(float(qr.grad.abs().max()), 0.05)
    assert float((dq.float().cpu() - qr.grad).abs().max()) / scale_q < _gtol(dtype)
    assert rel(dkv, kvr.grad) < _gtol(dtype)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('B,N,Nk,heads,D', [(1, 4096, 1024, 2, 64), (2, 1024, 1024, 1, 64), (1, 4096, 4096, 1, 64), (1, 2048, 4096, 5, 64),
                                            (1, 1000, 4100, 1, 32)])
def test_mit_attention_headline_key_counts(ops, dtype, B, N, Nk, heads, D):
    """Nk = 1024 is what every stage of the headline (configs[1], 1024^2) runs, Nk = 4096 what configs[4] (MiT-B5, 2048^2) runs:
    the key loop of the forward and both backward kernels over many key tiles (and a ragged one), against plain fp32 torch."""
    C = heads * D
    qq, kv, do = rnd(B * N, C), rnd(B * Nk, 2 * C, seed=1), rnd(B * N, C, seed=2)
    qr, kvr = q(qq, dtype).requires_grad_(True), q(kv, dtype).requires_grad_(True)
    qh = qr.view(B, N, heads, D).transpose(1, 2)
    k, v = kvr.view(B, Nk, 2, heads, D).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(qh @ k.transpose(-2, -1) * D ** -0.5, -1) @ v).transpose(1, 2).reshape(B * N, C)
    ref.backward(q(do, dtype))
    o, lse = ops.mit_attn_fwd(dev(qq, dtype), dev(kv, dtype), B, N, Nk, heads)
    assert rel(o, ref) < _tol(dtype)
    dq, dkv = ops.mit_attn_bwd(dev(qq, dtype), dev(kv, dtype), o, dev(do, dtype), lse, B, N, Nk, heads)
    assert rel(dq, qr.grad) < _gtol(dtype)
    assert rel(dkv, kvr.grad) < _gtol(dtype)


def test_mit_attention_online_softmax_rescale(ops):
    """a key far larger than the rest, placed in a late key tile, forces the running-max rescale branch"""
    B, N, Nk, heads, D = 1, 64, 200, 1, 64
    qq, kv = rnd(N, D), rnd(Nk, 2 * D, seed=1)
    kv[150, :D] = 6.0 * qq[5]
    qh, k, v = qq[None], kv[None, :, :D], kv[None, :, D:]
    ref = (torch.softmax(qh @ k.transpose(-2, -1) * D ** -0.5, -1) @ v)[0]
    o, _ = ops.mit_attn_fwd(dev(qq), dev(kv), B, N, Nk, heads)
    assert rel(o, ref) < 1e-4


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('C,k,s,pad', [(3, 7, 4, 3), (32, 3, 2, 1), (64, 8, 8, 0), (16, 2, 2, 0)])
def test_im2col_conv(ops, dtype, C, k, s, pad):
    B, H, W, O = 2, 32, 24, 40
    x, w = rnd(B, C, H, W), rnd(O, C, k, k, seed=1) / math.sqrt(C * k * k)
    xr, wr = q(x, dtype).requires_grad_(True), q(w, dtype).requires_grad_(True)
    ref = F.conv2d(xr, wr, stride=s, padding=pad)
    dy = rnd(*ref.shape, seed=2)
    ref.backward(q(dy, dtype))
    xn = ops.nchw_to_nhwc(dev(x), dtype)
    col, ho, wo = ops.im2col(xn, B, H, W, C, k, s, pad)
    wp, wpt = ops.pack_conv_weight(dev(w), dtype)
    y = ops.gemm_nt(col, wp)
    assert rel(y.view(B, ho, wo, O), ref.permute(0, 2, 3, 1)) < _tol(dtype)
    dyd = dev(dy.permute(0, 2, 3, 1).reshape(-1, O).contiguous(), dtype)
    gw = ops.unpack_conv_wgrad(ops.gemm_tn(dyd, col), O, C, k)
    assert rel(gw, wr.grad) < (1e-4 if dtype == torch.float32 else 1e-2)
    if C % 8 == 0:
        dcol = ops.gemm_nt(dyd, wpt)
        dx = ops.col2im(dcol, B, H, W, C, k, s, pad)
        assert rel(dx.view(B, H, W, C), xr.grad.permute(0, 2, 3, 1)) < _gtol(dtype)


@pytest.mark.parametrize('M,N,K,N2,act', [(1000, 256, 256, 8, 3), (4096 + 77, 256, 128, 32, 2), (300, 256, 320, 4, 0), (512, 192, 256, 8, 3), (256, 256, 64, 8, 3)])
def test_gemm_nt_nt_keeps_the_wide_activation_on_chip(ops, M, N, K, N2, act):
    """ss_gemm_nt_nt: act(a @ w1.T + b) @ w2.T in one kernel (a head's 1x1 conv + BN + act and its classifier, modules/heads.py:120-170,
    core/models.py:103-115) against the two GEMMs it replaces: the intermediate is rounded to bf16 at the same place, so only the fp32
    summation order of the second product differs.  Shapes the tile kernel does not take (N != 256, K < 128) run the two GEMMs."""
    T = torch.bfloat16
    a, w1, b, w2 = dev(rnd(M, K), T), dev(rnd(N, K, seed=1) / math.sqrt(K), T), dev(rnd(N, seed=2)), dev(rnd(N2, N, seed=3) / math.sqrt(N), T)
    out = ops.gemm_nt_nt(a, w1, b, act, w2)
    two = ops.gemm_nt(ops.gemm_nt(a, w1, bias=b, act=act), w2)
    assert out.shape == (M, N2) and rel(out, two) < 4e-3
    h = (a.float() @ w1.float().t() + b)
    h = {0: h, 2: h.clamp(min=0), 3: h.clamp(0, 6)}[act].to(T).float()
    assert rel(out, h @ w2.float().t()) < 6e-3


@pytest.mark.parametrize('B,H,W,C,s,O', [(2, 32, 24, 64, 8, 64), (3, 16, 40, 128, 4, 128), (2, 12, 20, 320, 2, 320), (1, 8, 8, 32, 2, 72),
                                         (2, 64, 64, 96, 2, 192), (5, 24, 8, 16, 4, 8)])
def test_conv_patch_gemms_without_patch_matrix(ops, B, H, W, C, s, O):
    """conv(kernel = stride = s) with the GEMM reading / writing the image itself (modules/segformer.py:27-29 sr conv, ConvNeXt
    down-sampling): forward, data gradient and weight / bias gradient against F.conv2d on the bf16-rounded operands (2e-2 budget, measured
    ~3e-3), and bit for bit against the im2col -> GEMM -> col2im route, which adds the same products in the same order."""
    dtype = torch.bfloat16
    x, w, b = rnd(B, C, H, W), rnd(O, C, s, s, seed=1) / math.sqrt(C * s * s), rnd(O, seed=3)
    xr, wr = q(x, dtype).requires_grad_(True), q(w, dtype).requires_grad_(True)
    ref = F.conv2d(xr, wr, b, stride=s)
    dy = rnd(*ref.shape, seed=2)
    ref.backward(q(dy, dtype))
    xn = ops.nchw_to_nhwc(dev(x), dtype)
    assert ops.conv_patch_ok(xn, B, H, W, C, s, O)
    wp, wpt = ops.pack_conv_weight(dev(w), dtype)
    y = ops.conv_patch_fwd(xn, B, H, W, C, s, wp, bias=dev(b))
    ho, wo = H // s, W // s
    assert rel(y.view(B, ho, wo, O), ref.permute(0, 2, 3, 1)) < 2e-2
    col, _, _ = ops.im2col(xn, B, H, W, C, s, s, 0)
    assert torch.equal(y, ops.gemm_nt(col, wp, bias=dev(b)))
    dyd = dev(dy.permute(0, 2, 3, 1).reshape(-1, O).contiguous(), dtype)
    gp, gb = ops.conv_patch_wgrad(dyd, xn, B, H, W, C, s)
    assert rel(ops.unpack_conv_wgrad(gp, O, C, s), wr.grad) < 1e-2
    assert rel(gb, q(dy, dtype).sum((0, 2, 3))) < 1e-4
    gp2, gb2 = ops.gemm_tn(dyd, col, colsum=True)
    assert rel(gp, gp2) < 1e-5 and rel(gb, gb2) < 1e-6            # the split plans may differ (tile choice), the products do not
    dx = ops.conv_patch_dgrad(dyd, wpt, B, H, W, C, s)
    assert rel(dx.view(B, H, W, C), xr.grad.permute(0, 2, 3, 1)) < 2e-2
    assert torch.equal(dx, ops.col2im(ops.gemm_nt(dyd, wpt), B, H, W, C, s, s, 0))


def test_conv_patch_ok_rejects_what_the_patch_gemm_cannot_address(ops):
    x = torch.zeros(2 * 32 * 32, 64, device='cuda', dtype=torch.bfloat16)
    assert ops.conv_patch_ok(x, 2, 32, 32, 64, 8, 64)
    assert not ops.conv_patch_ok(x.float(), 2, 32, 32, 64, 8, 64)                 # fp32 goes through im2col
    assert not ops.conv_patch_ok(x, 2, 32, 32, 64, 3, 64)                         # 32 % 3 != 0
    assert not ops.conv_patch_ok(x[: 2 * 32 * 32 // 4 * 4].view(-1, 16)[: 2 * 32 * 32], 2, 32, 32, 16, 2, 64)   # s*C = 32: not a multiple of 64
    with pytest.raises(ValueError):
        ops.conv_patch_fwd(x, 2, 32, 32, 64, 8, torch.zeros(64, 100, device='cuda', dtype=torch.bfloat16))


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('hi,wi,ho,wo,C', [(4, 4, 16, 16, 64), (8, 6, 16, 12, 16), (16, 16, 64, 64, 6), (5, 7, 13, 9, 8), (1, 1, 8, 8, 32),
                                           (32, 40, 128, 160, 6), (7, 9, 20, 31, 5), (16, 300, 8, 150, 6), (3, 3, 9, 9, 2), (1, 5, 4, 20, 6), (9, 300, 18, 600, 3), (2, 1, 8, 4, 8), (4, 300, 16, 1200, 6), (3, 257, 6, 514, 4)])
def test_bilinear(ops, dtype, hi, wi, ho, wo, C):
    B = 2
    x = rnd(B, C, hi, wi)
    xr = q(x, dtype).requires_grad_(True)
    ref = F.interpolate(xr, size=(ho, wo), mode='bilinear', align_corners=False)
    dy = rnd(*ref.shape, seed=1)
    ref.backward(q(dy, dtype))
    xd = dev(x.permute(0, 2, 3, 1).reshape(-1, C).contiguous(), dtype)
    wide = torch.zeros(B * ho * wo, C + 16, dtype=dtype, device='cuda')
    y = ops.bilinear_fwd(xd, B, hi, wi, ho, wo, out=wide[:, 8:8 + C])
    assert rel(y.view(B, ho, wo, C), ref.permute(0, 2, 3, 1)) < _tol(dtype)
    dyd = dev(dy.permute(0, 2, 3, 1).reshape(-1, C).contiguous(), dtype)
    dx = ops.bilinear_bwd(dyd, B, hi, wi, ho, wo)
    assert rel(dx.view(B, hi, wi, C), xr.grad.permute(0, 2, 3, 1)) < _tol(dtype)


@pytest.mark.parametrize('dtype', DTYPES)
def test_bilinear_two_sources(ops, dtype):
    """ss_bilinear_fwd2 = resize(a) + resize(b) of two map sizes in one pass, against torch and against the two-pass form"""
    B, C, ho, wo = 2, 24, 16, 24
    a, b = rnd(B, C, 4, 6), rnd(B, C, 8, 12, seed=1)
    ref = (F.interpolate(q(a, dtype), size=(ho, wo), mode='bilinear', align_corners=False)
           + F.interpolate(q(b, dtype), size=(ho, wo), mode='bilinear', align_corners=False)).permute(0, 2, 3, 1).reshape(-1, C)
    ad = dev(a.permute(0, 2, 3, 1).reshape(-1, C).contiguous(), dtype)
    bd = dev(b.permute(0, 2, 3, 1).reshape(-1, C).contiguous(), dtype)
    y = ops.bilinear_fwd2(ad, 4, 6, bd, 8, 12, B, ho, wo)
    assert rel(y, ref) < _tol(dtype)
    two = ops.bilinear_fwd(bd, B, 8, 12, ho, wo, out=ops.bilinear_fwd(ad, B, 4, 6, ho, wo), accumulate=True)
    assert rel(y, two) < (1e-6 if dtype == torch.float32 else 8e-3)
    with pytest.raises(ValueError):
        ops.bilinear_fwd2(ad[:, :20], 4, 6, bd[:, :20], 8, 12, B, ho, wo)


def test_small_helpers(ops):
    x = rnd(64, 32)
    xb = ops.cast(dev(x), torch.bfloat16)
    assert torch.equal(xb.cpu(), x.to(torch.bfloat16))
    assert torch.equal(ops.cast(xb, torch.float32).cpu(), x.to(torch.bfloat16).float())
    assert torch.equal(ops.transpose_cast(dev(x), torch.float32).cpu(), x.t().contiguous())
    assert torch.equal(ops.add(dev(x), dev(x)).cpu(), 2 * x)
    s = torch.tensor([0.0, 2.0])
    assert torch.equal(ops.rowscale(dev(x), dev(s), 32).cpu(), x * s.repeat_interleave(32)[:, None])
    out = torch.zeros(2 * 10, 48, device='cuda')
    ops.bcast_rows(dev(x[:2]), out[:, 8:40], 10)
    assert torch.equal(out[:, 8:40].cpu(), x[:2].repeat_interleave(10, 0))
    with pytest.raises((RuntimeError, ValueError)):
        ops.add(x, x)                      # CPU tensors are refused: no fallback path


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('B,H,W,inter,heads', [(2, 16, 24, 64, 4), (1, 8, 16, 128, 4), (2, 16, 16, 128, 2), (1, 16, 8, 32, 8), (1, 8, 8, 64, 64),
                                               (2, 16, 16, 256, 64), (1, 16, 8, 128, 64), (1, 8, 16, 128, 16), (1, 8, 8, 48, 12), (1, 8, 8, 40, 10)])
def test_lawin_window_attention(ops, dtype, B, H, W, inter, heads):
    """per (8x8 window, head) softmax(theta phi^T / sqrt(d)) g (modules/lawin.py:72-91) and its backward; in bf16 head dims 16/32/64
    take the one-head-per-wave MFMA kernels and 2/4/8 (embed 512: 64 heads of 4; embed 256: 64 heads of 2, 16 of 8) the grouped ones
    (16 channels = 16/d heads per wave); the rest (d = 1, inter % 16 != 0) and fp32 the vector-ALU kernels."""
    d = inter // heads
    nh, nw = H // 8, W // 8
    nwin = B * nh * nw
    theta, pg, dy = rnd(B * H * W, inter), rnd(nwin * 64, 2 * inter, seed=1), rnd(B * H * W, inter, seed=2)
    tr, pr = q(theta, dtype).requires_grad_(True), q(pg, dtype).requires_grad_(True)
    # image-ordered rows -> [window, head, 64, d]
    tw = tr.view(B, nh, 8, nw, 8, heads, d).permute(0, 1, 3, 5, 2, 4, 6).reshape(nwin, heads, 64, d)
    phi = pr[:, :inter].reshape(nwin, 64, heads, d).transpose(1, 2)
    g = pr[:, inter:].reshape(nwin, 64, heads, d).transpose(1, 2)
    yw = torch.softmax(tw @ phi.transpose(-1, -2) / math.sqrt(d), -1) @ g
    ref = yw.view(B, nh, nw, heads, 8, 8, d).permute(0, 1, 4, 2, 5, 3, 6).reshape(B * H * W, inter)
    ref.backward(q(dy, dtype))
    y, lse = ops.lawin_attn_fwd(dev(theta, dtype), dev(pg, dtype), B, H, W, heads)
    assert rel(y, ref) < _tol(dtype)
    dth, dpg = ops.lawin_attn_bwd(dev(theta, dtype), dev(pg, dtype), y, dev(dy, dtype), lse, B, H, W, heads)
    assert rel(dth, tr.grad) < _gtol(dtype)
    assert rel(dpg, pr.grad) < _gtol(dtype)
    # theta / y as channel slices of wider buffers (how LawinHead calls it)
    wide = dev(rnd(B * H * W, inter + 16, seed=4), dtype)
    y2, _ = ops.lawin_attn_fwd(wide[:, 8:8 + inter], dev(pg, dtype), B, H, W, heads)
    tw2 = wide[:, 8:8 + inter].float().cpu().view(B, nh, 8, nw, 8, heads, d).permute(0, 1, 3, 5, 2, 4, 6).reshape(nwin, heads, 64, d)
    ref2 = (torch.softmax(tw2 @ phi.detach().transpose(-1, -2) / math.sqrt(d), -1) @ g.detach()).view(B, nh, nw, heads, 8, 8, d)
    assert rel(y2, ref2.permute(0, 1, 4, 2, 5, 3, 6).reshape(B * H * W, inter)) < _tol(dtype)


def _gather_windows(G, B, H, W, r):
    """window-ordered [nwin*64, C] view of the pooled grid (the 8x8 cell block of every window), as the kernels gather it."""
    nh, nw, st = H // 8, W // 8, 8 // r
    gh, gw = (nh - 1) * st + 8, (nw - 1) * st + 8
    Gi = G.reshape(B, gh, gw, -1)
    rows = []
    for b in range(B):
        for i in range(nh):
            for j in range(nw):
                rows.append(Gi[b, i * st:i * st + 8, j * st:j * st + 8].reshape(64, -1))
    return torch.cat(rows, 0)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('C,heads,r', [(128, 16, 2), (128, 8, 4), (128, 4, 8), (256, 4, 2), (128, 1, 4), (64, 8, 2), (128, 64, 8)])
def test_lawin_position_mixing(ops, dtype, C, heads, r):
    """modules/lawin.py:44-46,62-69 -- out = x + W_h x + b_h over the 64 window positions, its transpose, and dW / db; head widths
    8..128 with C % 128 == 0 take the matrix-core kernels in bf16, everything else the vector-ALU ones."""
    B, H, W = 2, 16, 24
    nwin = B * (H // 8) * (W // 8)
    st = 8 // r
    gh, gw = (H // 8 - 1) * st + 8, (W // 8 - 1) * st + 8
    G = rnd(B * gh * gw, C)
    Wm, bm = rnd(heads, 64, 64, seed=1) / 8, 0.1 * rnd(heads, 64, seed=2)
    d = rnd(nwin * 64, C, seed=3)
    cg = C // heads
    xw = _gather_windows(q(G, dtype), B, H, W, r).reshape(nwin, 64, heads, cg)
    ref = xw + torch.einsum('hqp,wphc->wqhc', Wm, xw) + bm.t().reshape(1, 64, heads, 1)
    out = ops.lawin_mix(dev(Wm), dev(bm), B, H, W, r, heads, G=dev(G, dtype))
    tol = 1e-2 if dtype == torch.bfloat16 else 1e-4
    assert rel(out, ref.reshape(nwin * 64, C)) < tol
    dq = q(d, dtype).reshape(nwin, 64, heads, cg)
    refT = dq + torch.einsum('hpq,wphc->wqhc', Wm, dq)
    outT = ops.lawin_mix(dev(Wm), None, B, H, W, r, heads, xwin=dev(d, dtype), transpose=True)
    assert rel(outT, refT.reshape(nwin * 64, C)) < tol
    dWb = ops.lawin_mix_wgrad(dev(G, dtype), dev(d, dtype), B, H, W, r, heads)
    refW = torch.einsum('wqhc,wphc->hqp', dq, xw)
    assert rel(dWb[:, :, :64], refW) < tol
    assert rel(dWb[:, :, 64], dq.sum((0, 3)).t()) < tol


@pytest.mark.parametrize('dtype', DTYPES)
def test_droppath_keep_matches_torch_ops(ops, dtype):
    """modules/helpers.py:20-24 as torch runs it: kp + rand (rounded to the dtype), floor_, then / kp."""
    for kp in (0.9, 0.9667, 0.5, 1.0):
        u = torch.rand((4096, 1, 1), device='cuda').to(dtype)
        want = (kp + u).floor_().float() / kp
        got = ops.droppath_keep(u, kp)
        assert got.shape == (4096,) and torch.equal(got, want.reshape(-1))


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('k,C,H,W,pt,pl,act', [(7, 96, 19, 70, 3, 3, 0), (5, 32, 9, 130, 2, 2, 2), (3, 64, 8, 64, 1, 1, 3), (7, 32, 5, 3, 3, 3, 1),
                                                (5, 32, 12, 66, 1, 3, 0), (7, 40, 10, 20, 3, 3, 0), (7, 192, 150, 33, 3, 3, 0), (3, 96, 70, 129, 1, 1, 2),
                                                (5, 160, 3, 1, 2, 2, 3)])
@pytest.mark.parametrize('path', ['tile', 'mfma'])
def test_dwconvk_stride1_tiled_and_fallback(ops, dtype, k, C, H, W, pt, pl, act, path, monkeypatch):
    """depthwise k x k, stride 1: the LDS-tiled kernel (C % 32 == 0; ragged tiles, asymmetric padding, fused activation), the per-pixel
    fallback (C = 40) and, with SS_DWK_FWD=mfma (read per call), the 4x4x4-MFMA kernel (bf16, centred padding) against F.conv2d on
    explicitly padded input."""
    monkeypatch.setenv('SS_DWK_FWD', path)
    B = 2
    x, w, b = rnd(B, C, H, W), rnd(C, 1, k, k, seed=1) / k, rnd(C, seed=2)
    Ho, Wo = H + 2 * pt - k + 1 if pt * 2 >= k - 1 else H, W
    Ho, Wo = H, W                                              # 'same' output size: bottom / right padding takes the rest
    xp = F.pad(q(x, dtype), (pl, k - 1 - pl, pt, k - 1 - pt))
    ref = F.conv2d(xp, w, b, groups=C)
    ref = {0: ref, 1: F.gelu(ref), 2: F.relu(ref), 3: F.relu6(ref)}[act]
    xn = dev(x.permute(0, 2, 3, 1).reshape(-1, C).contiguous(), dtype)
    wt = dev(w.reshape(C, k * k).t().contiguous())
    y = ops.dwconvk_fwd(xn, wt, dev(b), B, H, W, k, 1, pt, pl, Ho, Wo, act=act)
    assert rel(y.view(B, Ho, Wo, C), ref.permute(0, 2, 3, 1)) < _tol(dtype)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('k,C,H,W,act', [(3, 64, 64, 128, 2), (5, 96, 38, 150, 2), (3, 192, 18, 70, 3), (5, 32, 9, 64, 0), (3, 40, 20, 66, 2), (5, 64, 10, 40, 2)])
def test_dwconvk_stride2_tiled_and_fallback(ops, dtype, k, C, H, W, act):
    """depthwise k x k, stride 2 with the asymmetric AutoPadding of modules/layers.py:146-164 (TopFormer's down-sampling blocks), odd
    sizes, with and without the fused residual, against F.conv2d on the explicitly padded input.  (An LDS-tiled stride-2 variant
    passed this test and was slower than the per-pixel kernel: 1.09 vs 0.85 ms per TopFormer-B step; not kept.)"""
    B = 2
    x, w, b = rnd(B, C, H, W), rnd(C, 1, k, k, seed=1) / k, rnd(C, seed=2)
    Ho, Wo = -(-H // 2), -(-W // 2)
    ph, pw = max((Ho - 1) * 2 + k - H, 0), max((Wo - 1) * 2 + k - W, 0)
    pt, pl = ph // 2, pw // 2
    xp = F.pad(q(x, dtype), (pl, pw - pl, pt, ph - pt))
    ref = F.conv2d(xp, w, b, stride=2, groups=C)
    ref = {0: ref, 2: F.relu(ref), 3: F.relu6(ref)}[act]
    assert ref.shape[-2:] == (Ho, Wo)
    xn = dev(x.permute(0, 2, 3, 1).reshape(-1, C).contiguous(), dtype)
    wt = dev(w.reshape(C, k * k).t().contiguous())
    y = ops.dwconvk_fwd(xn, wt, dev(b), B, H, W, k, 2, pt, pl, Ho, Wo, act=act)
    assert rel(y.view(B, Ho, Wo, C), ref.permute(0, 2, 3, 1)) < _tol(dtype)
    res = rnd(B * Ho * Wo, C, seed=3)
    y2 = ops.dwconvk_fwd(xn, wt, dev(b), B, H, W, k, 2, pt, pl, Ho, Wo, act=act, res=dev(res, dtype))
    assert rel(y2.view(B, Ho, Wo, C), ref.permute(0, 2, 3, 1) + q(res, dtype).view(B, Ho, Wo, C)) < _tol(dtype)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('k,C,H,W', [(7, 96, 19, 70), (5, 32, 9, 130), (3, 64, 8, 64), (7, 8, 5, 3), (7, 40, 33, 20), (7, 192, 70, 33), (5, 96, 40, 64),
                                     (3, 32, 3, 5), (7, 64, 1, 1), (3, 160, 33, 129)])
def test_dwconvk_weight_gradient(ops, dtype, k, C, H, W):
    """dW, db of a depthwise k x k (stride 1, padding k//2) against torch autograd; strips of several tiles, ragged edges; bitwise
    reproducible.  bf16 with C % 32 == 0 runs on the matrix cores (dwk_wgrad_mfma_kernel: 16 channels per 4x4x4 MFMA), the rest on the
    vector-ALU tile kernel."""
    B = 3
    x, g = rnd(B, C, H, W), rnd(B, C, H, W, seed=1)
    w = torch.zeros(C, 1, k, k, requires_grad=True)
    bias = torch.zeros(C, requires_grad=True)
    F.conv2d(q(x, dtype), w, bias, padding=k // 2, groups=C).backward(q(g, dtype))
    xn = dev(x.permute(0, 2, 3, 1).reshape(-1, C).contiguous(), dtype)
    gn = dev(g.permute(0, 2, 3, 1).reshape(-1, C).contiguous(), dtype)
    out = ops.dwconvk_wgrad(gn, xn, B, H, W, k)
    assert out.shape == (k * k + 1, C)
    assert rel(out[:k * k].t().reshape(C, 1, k, k), w.grad) < 1e-4 and rel(out[k * k], bias.grad) < 1e-4
    assert torch.equal(out, ops.dwconvk_wgrad(gn, xn, B, H, W, k))


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('M,N,K', [(1000, 384, 96), (70000, 768, 192), (513, 96, 384), (131072 + 5, 512, 256)])
def test_gemm_nt_gelu_gradient_epilogue(ops, dtype, M, N, K):
    """act 6: (A . B^T) * GELU'(res), the GELU backward of modules/convnext.py:43-45 inside the data-gradient GEMM."""
    a, b, z = rnd(M, K), rnd(N, K, seed=1) / math.sqrt(K), rnd(M, N, seed=2) * 2
    zr = q(z, dtype).requires_grad_(True)
    F.gelu(zr).backward(torch.ones_like(zr))
    ref = (q(a, dtype) @ q(b, dtype).t()) * zr.grad
    out = ops.gemm_nt(dev(a, dtype), dev(b, dtype), res=dev(z, dtype), act=ops.GELU_GRAD)
    assert rel(out, ref) < _tol(dtype)
    with pytest.raises(RuntimeError):
        ops.gemm_nt(dev(a, dtype), dev(b, dtype), act=ops.GELU_GRAD)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('shape', [(4096, 384, 96), (1000, 768, 192), (640, 3072, 768), (130, 24, 16), (131072 + 40, 512, 256)])
def test_gemm_nt_gelu2_equals_gemm_then_gelu(dtype, shape):
    """Linear + GELU with both outputs from one epilogue == the GEMM followed by the GELU kernel, bit for bit (every tile kernel)."""
    from semseg_amd import ops
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N)
    a = (torch.randn((M, K), generator=g)).to(dtype).cuda()
    w = (torch.randn((N, K), generator=g) * K ** -0.5).to(dtype).cuda()
    b = torch.randn(N, generator=g).cuda()
    z, act = ops.gemm_nt_gelu2(a, w, bias=b)
    z0 = ops.gemm_nt(a, w, bias=b)
    assert torch.equal(z, z0) and torch.equal(act, ops.gelu_fwd(z0))
    ref = torch.nn.functional.gelu(z0.float())
    assert float((act.float() - ref).abs().max()) < (2e-2 if dtype == torch.bfloat16 else 1e-5) * max(1.0, float(ref.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('geom', [(2, 24, 40, 64, 7, 1), (1, 9, 70, 32, 3, 1), (2, 16, 16, 24, 5, 2)])
@pytest.mark.parametrize('path', ['tile', 'mfma'])
def test_dwconvk_fwd_with_residual(dtype, geom, path, monkeypatch):
    """y = dw(x) + res in one kernel (tiled, MFMA and generic paths) vs the two-step result computed in fp32."""
    from semseg_amd import ops
    monkeypatch.setenv('SS_DWK_FWD', path)
    B, H, W, C, k, s = geom
    g = torch.Generator().manual_seed(C + k)
    x = torch.randn((B * H * W, C), generator=g).to(dtype).cuda()
    w = (torch.randn((k * k, C), generator=g) * 0.2).cuda()
    y0 = ops.dwconvk_fwd(x.float(), w, None, B, H, W, k, stride=s)
    res = torch.randn(y0.shape, generator=g).to(dtype).cuda()
    y = ops.dwconvk_fwd(x, w, None, B, H, W, k, stride=s, res=res)
    want = y0 + res.float()
    assert float((y.float() - want).abs().max()) < (2e-2 if dtype == torch.bfloat16 else 1e-5) * max(1.0, float(want.abs().max()))
