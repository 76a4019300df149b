"""Functional (non-autograd) wrappers over the C ABI of libsemseg_hip.so.

Every function validates shapes / dtypes / contiguity on the host, borrows the tensors' device pointers for one
asynchronous launch on the current torch HIP stream and returns torch tensors.  PyTorch is used for allocation only.
There is no fallback: tensors must live on a HIP device and the library must be built.
"""
import os

import torch
from ._lib import lib, check

F32, BF16 = 0, 1
_DT = {torch.float32: F32, torch.bfloat16: BF16}
_WS = {}
COLRED_BLOCKS = 1024


def dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f'semseg_amd: unsupported dtype {t.dtype}')


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(t):
    if not t.is_cuda:
        raise RuntimeError('semseg_amd: tensors must live on a HIP device (no CPU fallback)')
    return t.device


def _p(t):
    return None if t is None else t.data_ptr()


def workspace(device, nbytes):
    """scratch for the entry point about to be launched: one buffer per (device, stream) -- launches on one stream run in order, so
    they can share it; two streams (runtime.side_stream) must not"""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    t = _WS.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(int(nbytes), 64 << 20), dtype=torch.uint8, device=device)
        _WS[key] = t
    return t


def _mat(t, name):
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f'{name}: expected a 2-D tensor with unit inner stride, got {tuple(t.shape)} / {t.stride()}')
    return t.shape[0], t.shape[1], t.stride(0)


def _f32(t, name, n=None):
    if t is None:
        return
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError(f'{name}: expected contiguous float32')
    if n is not None and t.numel() != n:
        raise ValueError(f'{name}: expected {n} elements, got {t.numel()}')


# ------------------------------------------------------------------ GEMM
def gemm_nt(a, b, out=None, bias=None, res=None, rowscale=None, rows_per_scale=0, act=0, out_f32=False):
    """out[M,N] = (a[M,K] @ b[N,K]^T + bias) * rowscale[row // rows_per_scale] + res."""
    dev = _dev(a)
    M, K, lda = _mat(a, 'a')
    N, Kb, ldb = _mat(b, 'b')
    if K != Kb or a.dtype != b.dtype:
        raise ValueError(f'gemm_nt: a {tuple(a.shape)} {a.dtype} vs b {tuple(b.shape)} {b.dtype}')
    odt = torch.float32 if out_f32 else a.dtype
    if out is None:
        out = torch.empty((M, N), dtype=odt, device=dev)
    Mo, No, ldc = _mat(out, 'out')
    if (Mo, No) != (M, N) or out.dtype != odt:
        raise ValueError('gemm_nt: bad out')
    ldr = 0
    if res is not None:
        Mr, Nr, ldr = _mat(res, 'res')
        if (Mr, Nr) != (M, N) or res.dtype != odt:
            raise ValueError('gemm_nt: bad res')
    _f32(bias, 'bias', N)
    if rowscale is not None:
        _f32(rowscale, 'rowscale')
        if rows_per_scale <= 0 or rowscale.numel() * rows_per_scale < M:
            raise ValueError('gemm_nt: bad rowscale')
    check(lib().ss_gemm_nt(dt(a), int(out_f32), _p(a), lda, _p(b), ldb, _p(out), ldc, M, N, K, _p(bias), _p(res), ldr,
                           _p(rowscale), rows_per_scale, act, _stream()))
    return out


def gemm_nt_nt(a, w1, bias, act, w2):
    """act(a @ w1.T + bias) @ w2.T with the [M, 256] intermediate kept on chip (ss_gemm_nt_nt) where the tile kernel takes the shape,
    else two GEMMs.  w2: [N2, 256], N2 <= 32."""
    dev = _dev(a)
    M, K, lda = _mat(a, 'a')
    N, K1, ldb = _mat(w1, 'w1')
    N2, Nw, ldw2 = _mat(w2, 'w2')
    if K1 != K or Nw != N or w1.dtype != a.dtype or w2.dtype != a.dtype:
        raise ValueError('gemm_nt_nt: shapes / dtypes do not chain')
    _f32(bias, 'bias', N)
    if not lib().ss_gemm_nt_nt_ok(dt(a), M, N, K, N2):
        return gemm_nt(gemm_nt(a, w1, bias=bias, act=act), w2)
    out = torch.empty((M, N2), dtype=a.dtype, device=dev)
    check(lib().ss_gemm_nt_nt(dt(a), _p(a), lda, _p(w1), ldb, _p(bias), act, _p(w2), ldw2, _p(out), N2, M, N, K, N2, _stream()))
    return out


def gemm_tn(a, b, out=None, accumulate=False, colsum=False):
    """out[M,N] (fp32) (+)= a[R,M]^T @ b[R,N].  colsum=True -> (out, column sums of a [M] fp32) from the same pass."""
    dev = _dev(a)
    R, M, lda = _mat(a, 'a')
    Rb, N, ldb = _mat(b, 'b')
    if R != Rb or a.dtype != b.dtype:
        raise ValueError(f'gemm_tn: a {tuple(a.shape)} vs b {tuple(b.shape)}')
    if out is None:
        if accumulate:
            raise ValueError('gemm_tn: accumulate needs out')
        out = torch.empty((M, N), dtype=torch.float32, device=dev)
    Mo, No, ldc = _mat(out, 'out')
    if (Mo, No) != (M, N) or out.dtype != torch.float32:
        raise ValueError('gemm_tn: bad out')
    nbytes = lib().ss_gemm_tn_workspace(dt(a), M, N, R)
    ws = workspace(dev, nbytes)
    if not colsum:
        check(lib().ss_gemm_tn(dt(a), _p(a), lda, _p(b), ldb, _p(out), ldc, M, N, R, _p(ws), ws.numel(), int(accumulate), _stream()))
        return out
    if M % 8 or lda % 8:
        check(lib().ss_gemm_tn(dt(a), _p(a), lda, _p(b), ldb, _p(out), ldc, M, N, R, _p(ws), ws.numel(), int(accumulate), _stream()))
        return out, colsum_any(a)
    cs = torch.empty(M, dtype=torch.float32, device=dev)
    check(lib().ss_gemm_tn_colsum(dt(a), _p(a), lda, _p(b), ldb, _p(out), ldc, M, N, R, _p(ws), ws.numel(), int(accumulate), _p(cs), 0,
                                  _stream()))
    return out, cs


def gemm_tn_keep_ok(a, b, rows_per_keep):
    """whether gemm_tn_keep takes a [R, M], b [R, N] with one keep multiplier per rows_per_keep rows (else: rowscale + gemm_tn)"""
    if not a.is_cuda or a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16:
        return False
    R, M, lda = _mat(a, 'a')
    Rb, N, ldb = _mat(b, 'b')
    return R == Rb and bool(lib().ss_gemm_tn_keep_ok(dt(a), _p(a), lda, _p(b), ldb, M, N, R, rows_per_keep))


def gemm_tn_keep(a, b, rowkeep, rows_per_keep):
    """(dW [M, N], db [M]) = ((k * a)^T @ b, column sums of k * a) with k[r] = rowkeep[r // rows_per_keep] in {0, c}: the DropPath-scaled
    weight / bias gradient from the unscaled gradient a (k-tiles of dropped samples are skipped, the slabs carry c)."""
    dev = _dev(a)
    R, M, lda = _mat(a, 'a')
    Rb, N, ldb = _mat(b, 'b')
    _f32(rowkeep, 'rowkeep')
    if R != Rb or rowkeep.numel() * rows_per_keep != R:
        raise ValueError('gemm_tn_keep: shapes do not match')
    out = torch.empty((M, N), dtype=torch.float32, device=dev)
    cs = torch.empty(M, dtype=torch.float32, device=dev)
    ws = workspace(dev, lib().ss_gemm_tn_workspace(dt(a), M, N, R))
    check(lib().ss_gemm_tn_colsum_keep(dt(a), _p(a), lda, _p(b), ldb, _p(out), N, M, N, R, _p(ws), ws.numel(), 0, _p(cs), 0, _p(rowkeep),
                                       rows_per_keep, _stream()))
    return out, cs


# ------------------------------------------------------------------ norms / reductions
def layernorm_fwd(x, gamma, beta, eps=1e-5):
    dev = _dev(x)
    rows, C, ld = _mat(x, 'x')
    if ld != C:
        raise ValueError('layernorm_fwd: x must be dense')
    _f32(gamma, 'gamma', C), _f32(beta, 'beta', C)
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=dev)
    rstd = torch.empty(rows, dtype=torch.float32, device=dev)
    check(lib().ss_layernorm_fwd(dt(x), _p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, C, eps, _stream()))
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, res=None, dy2=None):
    """-> dx (+ res, the gradient of the residual path, when given), dgb [2,C] fp32 (dgamma, dbeta); dy2: a second gradient of the
    same output, summed with dy inside the kernel."""
    dev = _dev(x)
    rows, C, ld = _mat(x, 'x')
    if ld != C or dy.shape != x.shape or not dy.is_contiguous() or dy.dtype != x.dtype:
        raise ValueError('layernorm_bwd: x/dy must be dense and alike')
    _f32(gamma, 'gamma', C), _f32(mean, 'mean', rows), _f32(rstd, 'rstd', rows)
    if res is not None and (res.shape != x.shape or res.dtype != x.dtype or not res.is_contiguous()):
        raise ValueError('layernorm_bwd: res must be dense and like x')
    if dy2 is not None and (dy2.shape != x.shape or dy2.dtype != x.dtype or not dy2.is_contiguous()):
        raise ValueError('layernorm_bwd: dy2 must be dense and like x')
    dx = torch.empty_like(x)
    dgb = torch.empty((2, C), dtype=torch.float32, device=dev)
    ws = workspace(dev, 1024 * 2 * C * 4)
    check(lib().ss_layernorm_bwd(dt(x), _p(dy), _p(dy2) if dy2 is not None else None, _p(x), _p(mean), _p(rstd), _p(gamma), _p(res) if res is not None else None, _p(dx), _p(dgb),
                                 dgb.data_ptr() + C * 4,
                                 rows, C, _p(ws), ws.numel(), _stream()))
    return dx, dgb


def colsum(x, groups=1, scale=1.0, out=None, accumulate=False):
    """out[g, c] (+)= scale * sum over the g-th block of rows of x[:, c]  (fp32)."""
    dev = _dev(x)
    rows_total, C, ld = _mat(x, 'x')
    if rows_total % groups:
        raise ValueError('colsum: rows not divisible by groups')
    if out is None:
        out = torch.empty((groups, C), dtype=torch.float32, device=dev)
    _f32(out, 'out', groups * C)
    ws = workspace(dev, groups * COLRED_BLOCKS * C * 4)
    check(lib().ss_colsum(dt(x), _p(x), ld, rows_total // groups, C, groups, scale, _p(out), int(accumulate), _p(ws), ws.numel(), _stream()))
    return out


def bn_stats(x, eps=1e-5, momentum=0.1, running_mean=None, running_var=None, nbt=None):
    """Batch statistics of x[rows, C] -> stats [2,C] (mean, rstd); updates the running buffers in place."""
    dev = _dev(x)
    rows, C, ld = _mat(x, 'x')
    _f32(running_mean, 'running_mean', C), _f32(running_var, 'running_var', C)
    if nbt is not None and nbt.dtype != torch.int64:
        raise ValueError('bn_stats: num_batches_tracked must be int64')
    stats = torch.empty((2, C), dtype=torch.float32, device=dev)
    ws = workspace(dev, COLRED_BLOCKS * 2 * C * 4)
    check(lib().ss_bn_stats(dt(x), _p(x), ld, rows, C, eps, momentum, _p(stats), _p(running_mean), _p(running_var), _p(nbt),
                            _p(ws), ws.numel(), _stream()))
    return stats


def gemm_nt_bn_stats(a, b, res=None, eps=1e-5, momentum=0.1, running_mean=None, running_var=None, nbt=None, out=None):
    """z = a @ b.T (+ res) and bn_stats(z) from one pass where the bf16 tile kernels take the shape -> (z, stats [2, N])."""
    dev = _dev(a)
    M, K, lda = _mat(a, 'a')
    N, Kb, ldb = _mat(b, 'b')
    if K != Kb or a.dtype != b.dtype:
        raise ValueError(f'gemm_nt_bn_stats: a {tuple(a.shape)} {a.dtype} vs b {tuple(b.shape)} {b.dtype}')
    ldr = 0
    if res is not None:
        Mr, Nr, ldr = _mat(res, 'res')
        if (Mr, Nr) != (M, N) or res.dtype != a.dtype:
            raise ValueError('gemm_nt_bn_stats: bad res')
    _f32(running_mean, 'running_mean', N), _f32(running_var, 'running_var', N)
    if nbt is not None and nbt.dtype != torch.int64:
        raise ValueError('gemm_nt_bn_stats: num_batches_tracked must be int64')
    z = torch.empty((M, N), dtype=a.dtype, device=dev) if out is None else out
    Mo, No, ldc = _mat(z, 'out')
    if (Mo, No) != (M, N) or z.dtype != a.dtype:
        raise ValueError('gemm_nt_bn_stats: bad out')
    stats = torch.empty((2, N), dtype=torch.float32, device=dev)
    ws = workspace(dev, max(-(-M // 128), COLRED_BLOCKS) * 2 * N * 4)
    check(lib().ss_gemm_nt_bn_stats(dt(a), _p(a), lda, _p(b), ldb, _p(z), ldc, M, N, K, _p(res), ldr, eps, momentum, _p(stats), _p(running_mean),
                                    _p(running_var), _p(nbt), _p(ws), ws.numel(), _stream()))
    return z, stats


def bn_eval_stats(running_mean, running_var, eps=1e-5):
    C = running_mean.numel()
    _dev(running_mean)
    _f32(running_mean, 'running_mean', C), _f32(running_var, 'running_var', C)
    stats = torch.empty((2, C), dtype=torch.float32, device=running_mean.device)
    check(lib().ss_bn_eval_stats(_p(running_mean), _p(running_var), C, eps, _p(stats), _stream()))
    return stats


def bn_apply(z, stats, gamma, beta, act=0, res=None, mask=None, rows_per_image=0, out=None):
    dev = _dev(z)
    rows, C, ldz = _mat(z, 'z')
    _f32(stats, 'stats', 2 * C), _f32(gamma, 'gamma', C), _f32(beta, 'beta', C)
    if out is None:
        out = torch.empty((rows, C), dtype=z.dtype, device=dev)
    ro, Co, ldy = _mat(out, 'out')
    if (ro, Co) != (rows, C) or out.dtype != z.dtype:
        raise ValueError('bn_apply: bad out')
    ldr = 0
    if res is not None:
        rr, Cr, ldr = _mat(res, 'res')
        if (rr, Cr) != (rows, C) or res.dtype != z.dtype:
            raise ValueError('bn_apply: bad res')
    if mask is not None:
        _f32(mask, 'mask')
        if rows_per_image <= 0 or mask.numel() * rows_per_image != rows * C:
            raise ValueError('bn_apply: bad mask')
    check(lib().ss_bn_apply(dt(z), _p(z), ldz, _p(stats), _p(gamma), _p(beta), _p(res), ldr, _p(mask), rows_per_image, _p(out), ldy,
                            rows, C, act, _stream()))
    return out


def bn_bwd(dy, z, stats, gamma, beta, act=0, mask=None, rows_per_image=0, training=True, out=None):
    """-> dz, dgb [2,C] fp32 (dgamma, dbeta)."""
    dev = _dev(z)
    rows, C, ldz = _mat(z, 'z')
    rd, Cd, ldd = _mat(dy, 'dy')
    if (rd, Cd) != (rows, C) or dy.dtype != z.dtype:
        raise ValueError('bn_bwd: dy/z mismatch')
    _f32(stats, 'stats', 2 * C), _f32(gamma, 'gamma', C), _f32(beta, 'beta', C)
    if mask is not None:
        _f32(mask, 'mask')
        if rows_per_image <= 0 or mask.numel() * rows_per_image != rows * C:
            raise ValueError('bn_bwd: bad mask')
    if out is None:
        out = torch.empty((rows, C), dtype=z.dtype, device=dev)
    ro, Co, ldo = _mat(out, 'out')
    if (ro, Co) != (rows, C) or out.dtype != z.dtype:
        raise ValueError('bn_bwd: bad out')
    dgb = torch.empty((2, C), dtype=torch.float32, device=dev)
    ws = workspace(dev, (COLRED_BLOCKS * 2 * C + 2 * C) * 4)
    check(lib().ss_bn_bwd(dt(z), _p(dy), ldd, _p(z), ldz, _p(stats), _p(gamma), _p(beta), _p(mask), rows_per_image, int(training),
                          _p(out), ldo, _p(dgb), rows, C, act, _p(ws), ws.numel(), _stream()))
    return out, dgb


# ------------------------------------------------------------------ depthwise conv
def mixffn_ok(x, Ch):
    return x.is_cuda and bool(lib().ss_mixffn_ok(dt(x), x.shape[1], Ch))


def mixffn_fwd(x, w1, b1, w9c, bd, w2, b2, B, H, W, res=None, keep=None, save_hidden=False):
    """(fc2(GELU(dw3x3(fc1(x) + b1) + bd)) + b2) * keep[image] + res in one kernel -> out [rows, C] (, h1, a2 [rows, Ch] if save_hidden)."""
    dev = _dev(x)
    rows, C, ld = _mat(x, 'x')
    Ch = w1.shape[0]
    if ld != C or rows != B * H * W or not mixffn_ok(x, Ch):
        raise ValueError(f'mixffn_fwd: x must be dense bf16 [B*H*W, C] with C in (64, 128) and hidden width 4C (C={C}, Ch={Ch})')
    for name, t, shape in (('w1', w1, (Ch, C)), ('w2', w2, (C, Ch))):
        if tuple(t.shape) != shape or t.dtype != x.dtype or not t.is_contiguous():
            raise ValueError(f'mixffn_fwd: {name} must be contiguous {shape} in the compute dtype')
    _f32(b1, 'b1', Ch), _f32(bd, 'bd', Ch), _f32(b2, 'b2', C), _f32(w9c, 'w9c', 9 * Ch)
    if res is not None and (res.shape != x.shape or res.dtype != x.dtype or not res.is_contiguous()):
        raise ValueError('mixffn_fwd: res must match x')
    if keep is not None:
        _f32(keep, 'keep', B)
    out = torch.empty_like(x)
    h1 = a2 = None
    if save_hidden:
        h1 = torch.empty((rows, Ch), dtype=x.dtype, device=dev)
        a2 = torch.empty((rows, Ch), dtype=x.dtype, device=dev)
    check(lib().ss_mixffn_fwd(dt(x), _p(x), _p(w1), _p(b1), _p(w9c), _p(bd), _p(w2), _p(b2), _p(res), _p(keep), _p(out), _p(h1), _p(a2),
                              B, H, W, C, Ch, _stream()))
    return (out, h1, a2) if save_hidden else out


def dwconv3_fwd(x, w9c, bias, B, H, W, act=0):
    dev = _dev(x)
    rows, C, ld = _mat(x, 'x')
    if ld != C or rows != B * H * W:
        raise ValueError('dwconv3_fwd: x must be dense [B*H*W, C]')
    _f32(w9c, 'w9c', 9 * C), _f32(bias, 'bias', C)
    y = torch.empty_like(x)
    check(lib().ss_dwconv3_fwd(dt(x), _p(x), _p(w9c), _p(bias), _p(y), B, H, W, C, act, _stream()))
    return y


def dwconv3_bwd(dy, x, w9c, bias, B, H, W, act=0):
    """-> g = dy * act'(dw(x)+b) [rows, C], dw_db [10, C] fp32 (9 taps tap-major, then bias)."""
    dev = _dev(x)
    rows, C, ld = _mat(x, 'x')
    if ld != C or rows != B * H * W or dy.shape != x.shape or not dy.is_contiguous() or dy.dtype != x.dtype:
        raise ValueError('dwconv3_bwd: bad x / dy')
    _f32(w9c, 'w9c', 9 * C), _f32(bias, 'bias', C)
    g = torch.empty_like(x)
    dwdb = torch.empty((10, C), dtype=torch.float32, device=dev)
    ws = workspace(dev, COLRED_BLOCKS * 10 * C * 4)
    check(lib().ss_dwconv3_bwd(dt(x), _p(dy), _p(x), _p(w9c), _p(bias), _p(g), _p(dwdb), B, H, W, C, act, _p(ws), ws.numel(), _stream()))
    return g, dwdb


def dwconv3_bwd_fused(dy, x, w9c, bias, B, H, W, act=0):
    """-> dx [rows, C] (gradient of the depthwise conv's input), dw_db [10, C] fp32 (9 taps tap-major, then bias): one pass."""
    dev = _dev(x)
    rows, C, ld = _mat(x, 'x')
    if ld != C or rows != B * H * W or dy.shape != x.shape or not dy.is_contiguous() or dy.dtype != x.dtype:
        raise ValueError('dwconv3_bwd_fused: bad x / dy')
    _f32(w9c, 'w9c', 9 * C), _f32(bias, 'bias', C)
    dx = torch.empty_like(x)
    dwdb = torch.empty((10, C), dtype=torch.float32, device=dev)
    ws = workspace(dev, COLRED_BLOCKS * 10 * C * 4)
    check(lib().ss_dwconv3_bwd_fused(dt(x), _p(dy), _p(x), _p(w9c), _p(bias), _p(dx), _p(dwdb), B, H, W, C, act, _p(ws), ws.numel(), _stream()))
    return dx, dwdb


def dwconv3_lane_ok(x):
    """True where ss_dwconv3_bwd_fused runs its lane-per-column kernel (dwl_ok in dwconv.hip)."""
    return bool(x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] % 64 == 0 and os.environ.get('SS_DW_LANE', '1')[:1] != '0')


def dwconv3_bwd_fc2_ok(dout, x, B, H, W):
    return bool(x.is_cuda and x.dtype == torch.bfloat16 and dout.dtype == x.dtype and dout.dim() == 2 and x.dim() == 2 and dout.is_contiguous()
                and x.is_contiguous() and lib().ss_dwconv3_bwd_fc2_ok(dt(x), B, H, W, x.shape[1], dout.shape[1]))


def dwconv3_bwd_fc2(dout, w2t, keep, x, w9c, bias, B, H, W):
    """Mix-FFN backward from the gradient of fc2's output to the gradient of fc1's output: da2 = (dout @ w2t.T) * keep[image] stays on
    chip.  dout [rows, K], w2t [C, K], x = fc1's output [rows, C].  -> dx [rows, C], dw_db [10, C] fp32."""
    dev = _dev(x)
    rows, C, ld = _mat(x, 'x')
    K = dout.shape[1]
    if ld != C or rows != B * H * W or dout.shape[0] != rows or tuple(w2t.shape) != (C, K) or w2t.dtype != x.dtype or not w2t.is_contiguous():
        raise ValueError('dwconv3_bwd_fc2: bad x / dout / w2t')
    _f32(w9c, 'w9c', 9 * C), _f32(bias, 'bias', C)
    if keep is not None:
        _f32(keep, 'keep', B)
    dx = torch.empty_like(x)
    dwdb = torch.empty((10, C), dtype=torch.float32, device=dev)
    ws = workspace(dev, COLRED_BLOCKS * 10 * C * 4)
    check(lib().ss_dwconv3_bwd_fc2(dt(x), _p(dout), _p(w2t), _p(keep), _p(x), _p(w9c), _p(bias), _p(dx), _p(dwdb), B, H, W, C, K,
                                   _p(ws), ws.numel(), _stream()))
    return dx, dwdb


# ------------------------------------------------------------------ MiT attention
def mit_attn_fwd(q, kv, B, N, Nk, heads, scale=None):
    """q [B*N, C], kv [B*Nk, 2C] -> o [B*N, C], lse [B, heads, round32(N)]."""
    dev = _dev(q)
    rows, C, ld = _mat(q, 'q')
    rk, C2, ldk = _mat(kv, 'kv')
    D = C // heads
    if ld != C or ldk != C2 or rows != B * N or rk != B * Nk or C2 != 2 * C or kv.dtype != q.dtype or D * heads != C:
        raise ValueError(f'mit_attn_fwd: bad shapes q {tuple(q.shape)} kv {tuple(kv.shape)}')
    o = torch.empty_like(q)
    lse = torch.empty((B, heads, (N + 31) // 32 * 32), dtype=torch.float32, device=dev)
    nbytes = lib().ss_mit_attn_fwd_workspace(dt(q), B, Nk, heads, D)
    ws = workspace(dev, nbytes)
    check(lib().ss_mit_attn_fwd(dt(q), _p(q), _p(kv), _p(o), _p(lse), B, N, Nk, heads, D, float(D) ** -0.5 if scale is None else float(scale),
                                _p(ws), ws.numel(), _stream()))
    return o, lse


def mit_attn_bwd(q, kv, o, dout, lse, B, N, Nk, heads, scale=None):
    dev = _dev(q)
    rows, C, ld = _mat(q, 'q')
    D = C // heads
    for t, nm in ((o, 'o'), (dout, 'dout')):
        if t.shape != q.shape or t.dtype != q.dtype or not t.is_contiguous():
            raise ValueError(f'mit_attn_bwd: bad {nm}')
    if not q.is_contiguous() or not kv.is_contiguous() or kv.shape != (B * Nk, 2 * C) or rows != B * N:
        raise ValueError('mit_attn_bwd: bad q / kv')
    _f32(lse, 'lse', B * heads * ((N + 31) // 32 * 32))
    dq = torch.empty_like(q)
    dkv = torch.empty_like(kv)
    nbytes = lib().ss_mit_attn_bwd_workspace(dt(q), B, N, Nk, heads, D)
    ws = workspace(dev, nbytes)
    check(lib().ss_mit_attn_bwd(dt(q), _p(q), _p(kv), _p(o), _p(dout), _p(lse), _p(dq), _p(dkv), B, N, Nk, heads, D,
                                float(D) ** -0.5 if scale is None else float(scale), _p(ws), ws.numel(), _stream()))
    return dq, dkv


# ------------------------------------------------------------------ im2col / col2im / bilinear
def conv_out_size(n, k, s, pad):
    return (n + 2 * pad - k) // s + 1


def im2col(x, B, H, W, C, k, s, pad):
    """x [B*H*W, C] -> col [B*ho*wo, Kp], Kp = round8(k*k*C); column order (ky, kx, c)."""
    dev = _dev(x)
    if not x.is_contiguous() or x.numel() != B * H * W * C:
        raise ValueError('im2col: bad x')
    ho, wo = conv_out_size(H, k, s, pad), conv_out_size(W, k, s, pad)
    Kp = (k * k * C + 7) // 8 * 8
    col = torch.empty((B * ho * wo, Kp), dtype=x.dtype, device=dev)
    check(lib().ss_im2col(dt(x), _p(x), _p(col), B, H, W, C, k, s, pad, Kp, _stream()))
    return col, ho, wo


def col2im(col, B, H, W, C, k, s, pad):
    dev = _dev(col)
    ho, wo = conv_out_size(H, k, s, pad), conv_out_size(W, k, s, pad)
    if not col.is_contiguous() or col.shape != (B * ho * wo, k * k * C):
        raise ValueError('col2im: bad col')
    dx = torch.empty((B * H * W, C), dtype=col.dtype, device=dev)
    check(lib().ss_col2im(dt(col), _p(col), _p(dx), B, H, W, C, k, s, pad, _stream()))
    return dx


def conv_patch_ok(x, B, H, W, C, s, N):
    """True when conv(kernel = stride = s) on x [B*H*W, C] runs without the patch matrix (ss_conv_patch_*)."""
    return bool(x.is_cuda and x.is_contiguous() and x.numel() == B * H * W * C and lib().ss_conv_patch_ok(dt(x), B, H, W, C, s, N))


def conv_patch_fwd(x, B, H, W, C, s, w, bias=None):
    """x [B*H*W, C], w [N, s*s*C] (columns ky, kx, c) -> y [B*(H/s)*(W/s), N] = conv(x) + bias, no im2col matrix."""
    dev = _dev(x)
    N, K, ldw = _mat(w, 'w')
    if K != s * s * C or w.dtype != x.dtype or not conv_patch_ok(x, B, H, W, C, s, N):
        raise ValueError('conv_patch_fwd: shape outside the patch GEMM (see ss_conv_patch_ok)')
    _f32(bias, 'bias', N)
    y = torch.empty((B * (H // s) * (W // s), N), dtype=x.dtype, device=dev)
    check(lib().ss_conv_patch_fwd(dt(x), _p(x), B, H, W, C, s, _p(w), ldw, _p(bias), _p(y), N, N, _stream()))
    return y


def conv_patch_dgrad(dy, wt, B, H, W, C, s):
    """dy [M, N], wt [s*s*C, N] -> dx [B*H*W, C] (every pixel belongs to exactly one patch)."""
    dev = _dev(dy)
    M, N, lddy = _mat(dy, 'dy')
    K, Nw, ldw = _mat(wt, 'wt')
    if K != s * s * C or Nw != N or wt.dtype != dy.dtype or M != B * (H // s) * (W // s):
        raise ValueError('conv_patch_dgrad: bad shapes')
    dx = torch.empty((B * H * W, C), dtype=dy.dtype, device=dev)
    if not conv_patch_ok(dx, B, H, W, C, s, N):
        raise ValueError('conv_patch_dgrad: shape outside the patch GEMM (see ss_conv_patch_ok)')
    check(lib().ss_conv_patch_dgrad(dt(dy), _p(dy), lddy, _p(wt), ldw, _p(dx), B, H, W, C, s, N, _stream()))
    return dx


def conv_patch_wgrad(dy, x, B, H, W, C, s, want_bias=True):
    """-> (dw [N, s*s*C] fp32, dbias [N] fp32 or None)."""
    dev = _dev(dy)
    M, N, lddy = _mat(dy, 'dy')
    if M != B * (H // s) * (W // s) or x.dtype != dy.dtype or not conv_patch_ok(x, B, H, W, C, s, N):
        raise ValueError('conv_patch_wgrad: bad shapes')
    dw = torch.empty((N, s * s * C), dtype=torch.float32, device=dev)
    db = torch.empty(N, dtype=torch.float32, device=dev) if want_bias else None
    ws = workspace(dev, lib().ss_conv_patch_wgrad_workspace(dt(dy), B, H, W, C, s, N))
    check(lib().ss_conv_patch_wgrad(dt(dy), _p(dy), lddy, _p(x), B, H, W, C, s, N, _p(dw), s * s * C, _p(db), _p(ws), ws.numel(), _stream()))
    return dw, db


def bilinear_fwd2(x, hi, wi, x2, hi2, wi2, B, ho, wo):
    """resize(x [B*hi*wi, C]) + resize(x2 [B*hi2*wi2, C]) -> [B*ho*wo, C], summed in fp32 and rounded once"""
    dev = _dev(x)
    r, C, ldx = _mat(x, 'x')
    r2, C2, ldx2 = _mat(x2, 'x2')
    if r != B * hi * wi or r2 != B * hi2 * wi2 or C != C2 or x.dtype != x2.dtype or C % 8:
        raise ValueError('bilinear_fwd2: shapes do not match (C must be a multiple of 8)')
    out = torch.empty((B * ho * wo, C), dtype=x.dtype, device=dev)
    check(lib().ss_bilinear_fwd2(dt(x), _p(x), ldx, hi, wi, _p(x2), ldx2, hi2, wi2, _p(out), C, B, ho, wo, C, _stream()))
    return out


def bilinear_fwd(x, B, hi, wi, ho, wo, out=None, accumulate=False):
    """x [B*hi*wi, C] (row stride ld) -> out [B*ho*wo, C] (may be a channel slice of a wider buffer); accumulate: out += resize(x)."""
    dev = _dev(x)
    rows, C, ldx = _mat(x, 'x')
    if rows != B * hi * wi:
        raise ValueError('bilinear_fwd: bad x')
    if out is None:
        out = torch.empty((B * ho * wo, C), dtype=x.dtype, device=dev)
    ro, Co, ldy = _mat(out, 'out')
    if (ro, Co) != (B * ho * wo, C) or out.dtype != x.dtype:
        raise ValueError('bilinear_fwd: bad out')
    if accumulate:
        check(lib().ss_bilinear_fwd_add(dt(x), _p(x), ldx, _p(out), ldy, B, hi, wi, ho, wo, C, _stream()))
    else:
        check(lib().ss_bilinear_fwd(dt(x), _p(x), ldx, _p(out), ldy, B, hi, wi, ho, wo, C, _stream()))
    return out


def bilinear_bwd(dy, B, hi, wi, ho, wo, out=None, accumulate=False):
    """dy [B*ho*wo, C] -> dx [B*hi*wi, C]."""
    dev = _dev(dy)
    rows, C, ldy = _mat(dy, 'dy')
    if rows != B * ho * wo:
        raise ValueError('bilinear_bwd: bad dy')
    if out is None:
        if accumulate:
            raise ValueError('bilinear_bwd: accumulate needs out')
        out = torch.empty((B * hi * wi, C), dtype=dy.dtype, device=dev)
    ro, Co, ldx = _mat(out, 'out')
    if (ro, Co) != (B * hi * wi, C) or out.dtype != dy.dtype:
        raise ValueError('bilinear_bwd: bad out')
    check(lib().ss_bilinear_bwd(dt(dy), _p(dy), ldy, _p(out), ldx, B, hi, wi, ho, wo, C, int(accumulate), _stream()))
    return out


# ------------------------------------------------------------------ BatchNorm of a bilinear resize (the resized tensor never exists)
def _up_args(x, B, hi, wi, name):
    rows, C, ldx = _mat(x, 'x')
    if rows != B * hi * wi or C % 8:
        raise ValueError(f'{name}: x must be [B*hi*wi, C] with C % 8 == 0')
    return C, ldx


def _bvec(bvec, B, C, hi, wi, ho, wo):
    """optional per-image constant [B, C] fp32 added to the resized map (x2 resize only)"""
    if bvec is None:
        return
    _f32(bvec, 'bvec', B * C)
    if (ho, wo) != (2 * hi, 2 * wi):
        raise ValueError('bn_*_up: bvec is only supported for the x2 resize')


def bn_stats_up(x, B, hi, wi, ho, wo, eps=1e-5, momentum=0.1, running_mean=None, running_var=None, nbt=None, bvec=None):
    """bn_stats(bilinear_fwd(x, B, hi, wi, ho, wo)) without the resized tensor -> stats [2,C]."""
    dev = _dev(x)
    C, ldx = _up_args(x, B, hi, wi, 'bn_stats_up')
    _bvec(bvec, B, C, hi, wi, ho, wo)
    _f32(running_mean, 'running_mean', C), _f32(running_var, 'running_var', C)
    if nbt is not None and nbt.dtype != torch.int64:
        raise ValueError('bn_stats_up: num_batches_tracked must be int64')
    stats = torch.empty((2, C), dtype=torch.float32, device=dev)
    ws = workspace(dev, COLRED_BLOCKS * 2 * C * 4)
    check(lib().ss_bn_stats_up(dt(x), _p(x), ldx, _p(bvec), B, hi, wi, ho, wo, C, eps, momentum, _p(stats), _p(running_mean), _p(running_var), _p(nbt),
                               _p(ws), ws.numel(), _stream()))
    return stats


def bn_apply_up(x, B, hi, wi, ho, wo, stats, gamma, beta, act=0, out=None, bvec=None):
    """bn_apply(bilinear_fwd(x, ...), stats, gamma, beta, act) -> [B*ho*wo, C] (may be a channel slice of a wider buffer)."""
    dev = _dev(x)
    C, ldx = _up_args(x, B, hi, wi, 'bn_apply_up')
    _bvec(bvec, B, C, hi, wi, ho, wo)
    _f32(stats, 'stats', 2 * C), _f32(gamma, 'gamma', C), _f32(beta, 'beta', C)
    if out is None:
        out = torch.empty((B * ho * wo, C), dtype=x.dtype, device=dev)
    ro, Co, ldy = _mat(out, 'out')
    if (ro, Co) != (B * ho * wo, C) or out.dtype != x.dtype:
        raise ValueError('bn_apply_up: bad out')
    check(lib().ss_bn_apply_up(dt(x), _p(x), ldx, _p(bvec), B, hi, wi, ho, wo, _p(stats), _p(gamma), _p(beta), _p(out), ldy, C, act, _stream()))
    return out


def bn_bwd_up(dy, x, B, hi, wi, ho, wo, stats, gamma, beta, act=0, training=True, out=None, bvec=None):
    """bn_bwd(dy, bilinear_fwd(x, ...), ...) -> dz [B*ho*wo, C] (gradient of the RESIZED map), dgb [2,C]."""
    dev = _dev(x)
    C, ldx = _up_args(x, B, hi, wi, 'bn_bwd_up')
    _bvec(bvec, B, C, hi, wi, ho, wo)
    rd, Cd, ldd = _mat(dy, 'dy')
    if (rd, Cd) != (B * ho * wo, C) or dy.dtype != x.dtype:
        raise ValueError('bn_bwd_up: dy mismatch')
    _f32(stats, 'stats', 2 * C), _f32(gamma, 'gamma', C), _f32(beta, 'beta', C)
    if out is None:
        out = torch.empty((B * ho * wo, C), dtype=x.dtype, device=dev)
    ro, Co, ldo = _mat(out, 'out')
    if (ro, Co) != (B * ho * wo, C) or out.dtype != x.dtype:
        raise ValueError('bn_bwd_up: bad out')
    dgb = torch.empty((2, C), dtype=torch.float32, device=dev)
    ws = workspace(dev, (COLRED_BLOCKS * 2 * C + 2 * C) * 4)
    check(lib().ss_bn_bwd_up(dt(x), _p(dy), ldd, _p(x), ldx, _p(bvec), B, hi, wi, ho, wo, _p(stats), _p(gamma), _p(beta), int(training),
                             _p(out), ldo, _p(dgb), C, act, _p(ws), ws.numel(), _stream()))
    return out, dgb


def bn_bwd_up_to_input(dy, x, B, hi, wi, ho, wo, stats, gamma, beta, act=0, training=True, bvec=None, overwrite_dy=False):
    """bilinear_bwd(bn_bwd_up(...)[0]) -> dx [B*hi*wi, C], dgb [2,C]; for the x2 resize the gradient of the resized map is never written.
    overwrite_dy: the caller is done with dy -- the first pass may leave dy * act'(.) in it for the second pass (x2 resize, act = 1)."""
    if (ho, wo) != (2 * hi, 2 * wi):
        dz, dgb = bn_bwd_up(dy, x, B, hi, wi, ho, wo, stats, gamma, beta, act=act, training=training, bvec=bvec)
        return bilinear_bwd(dz, B, hi, wi, ho, wo), dgb
    dev = _dev(x)
    C, ldx = _up_args(x, B, hi, wi, 'bn_bwd_up_to_input')
    _bvec(bvec, B, C, hi, wi, ho, wo)
    rd, Cd, ldd = _mat(dy, 'dy')
    if (rd, Cd) != (B * ho * wo, C) or dy.dtype != x.dtype:
        raise ValueError('bn_bwd_up_to_input: dy mismatch')
    _f32(stats, 'stats', 2 * C), _f32(gamma, 'gamma', C), _f32(beta, 'beta', C)
    dx = torch.empty((B * hi * wi, C), dtype=x.dtype, device=dev)
    dgb = torch.empty((2, C), dtype=torch.float32, device=dev)
    ws = workspace(dev, (COLRED_BLOCKS * 2 * C + 2 * C) * 4)
    check(lib().ss_bn_bwd_up2_dx(dt(x), _p(dy), ldd, _p(x), ldx, _p(bvec), B, hi, wi, _p(stats), _p(gamma), _p(beta), int(training),
                                 _p(dx), C, _p(dgb), C, act, _p(ws), ws.numel(), _p(dy) if overwrite_dy else None, ldd if overwrite_dy else 0,
                                 _stream()))
    return dx, dgb


# ------------------------------------------------------------------ LawinHead: branch BatchNorms folded into the `cat` conv
def lawin_cat_fold(W, stats, gammas, betas, E, dtype):
    """W: cat conv weight [E, >= 4E] fp32 (row stride = W.stride(0)); stats / gammas / betas: three each (branches r = 8, 4, 2).
    -> (folded weight [E, 4E] in dtype, Wt [4E, E] = [(W_0+W_1+W_2+W_3)^T; W_1^T; W_2^T; W_3^T] in dtype, constant [E] fp32)."""
    dev = _dev(W)
    rows, cols, ldw = _mat(W, 'W')
    if W.dtype != torch.float32 or rows != E or cols < 4 * E:
        raise ValueError('lawin_cat_fold: W must be fp32 [E, >= 4E]')
    for t in list(stats) + list(gammas) + list(betas):
        _f32(t, 'stats/gamma/beta')
    Wo = torch.empty((E, 4 * E), dtype=dtype, device=dev)
    Wt = torch.empty((4 * E, E), dtype=dtype, device=dev)
    cc = torch.empty(E, dtype=torch.float32, device=dev)
    check(lib().ss_lawin_cat_fold(_DT[dtype], _p(W), ldw, *[_p(t) for t in stats], *[_p(t) for t in gammas], *[_p(t) for t in betas], E,
                                  _p(Wo), _p(Wt), _p(cc), _stream()))
    return Wo, Wt, cc


def lawin_branch_bn_sums(G, W, stats, dvb, B, E, rows, training):
    """-> dgb [3, 2, E] (dgamma, dbeta of the three branch BatchNorms), coef [3, 2, E] for bn_bwd_coef; G before lawin_cat_wgrad_fix."""
    dev = _dev(G)
    _, _, ldg = _mat(G, 'G')
    _, _, ldw = _mat(W, 'W')
    if G.dtype != torch.float32 or W.dtype != torch.float32:
        raise ValueError('lawin_branch_bn_sums: G and W must be fp32')
    _f32(dvb, 'dvb', B * E)
    dgb = torch.empty((3, 2, E), dtype=torch.float32, device=dev)
    coef = torch.empty((3, 2, E), dtype=torch.float32, device=dev)
    check(lib().ss_lawin_branch_bn_sums(_p(G), ldg, _p(W), ldw, *[_p(t) for t in stats], _p(dvb), B, E, rows, int(training), _p(dgb), _p(coef),
                                        _stream()))
    return dgb, coef


def bn_bwd_coef(dy, z, stats, gamma, beta, coef):
    """second pass of bn_bwd (act 0, no mask) with given first-pass means coef [2, C] -> dz."""
    dev = _dev(z)
    rows, C, ldz = _mat(z, 'z')
    rd, Cd, ldd = _mat(dy, 'dy')
    if (rd, Cd) != (rows, C) or dy.dtype != z.dtype:
        raise ValueError('bn_bwd_coef: dy/z mismatch')
    _f32(stats, 'stats', 2 * C), _f32(gamma, 'gamma', C), _f32(beta, 'beta', C), _f32(coef, 'coef', 2 * C)
    dz = torch.empty((rows, C), dtype=z.dtype, device=dev)
    check(lib().ss_bn_bwd_coef(dt(z), _p(dy), ldd, _p(z), ldz, _p(stats), _p(gamma), _p(beta), _p(coef), _p(dz), C, rows, C, _stream()))
    return dz


def lawin_cat_wgrad_fix(G, stats, gammas, betas, dvb, B, E):
    """in place on the first 4E columns of G [E, ld]: gradient of the folded GEMM's weight -> gradient of the cat conv weight."""
    _dev(G)
    rows, cols, ldg = _mat(G, 'G')
    if G.dtype != torch.float32 or rows != E or cols < 4 * E:
        raise ValueError('lawin_cat_wgrad_fix: G must be fp32 [E, >= 4E]')
    _f32(dvb, 'dvb', B * E)
    check(lib().ss_lawin_cat_wgrad_fix(_p(G), ldg, *[_p(t) for t in stats], *[_p(t) for t in gammas], *[_p(t) for t in betas], _p(dvb), B, E,
                                       _stream()))
    return G


# ------------------------------------------------------------------ BatchNorm + Mish + Dropout2d + narrow classifier in one pass
def bn_classify_ok(C, nc):
    ch = C // 8
    return C % 8 == 0 and 1 <= ch <= 64 and ch & (ch - 1) == 0 and 1 <= nc <= 8


def _cls_args(z, wp, mask, rows_per_image, name):
    rows, C, ldz = _mat(z, 'z')
    nc = wp.shape[0]
    _f32(wp, 'wp', nc * C)
    if not bn_classify_ok(C, nc):
        raise ValueError(f'{name}: C must be 8 * a power of two <= 512 and nc <= 8 (got C={C}, nc={nc})')
    if mask is not None:
        _f32(mask, 'mask')
        if rows_per_image <= 0 or mask.numel() * rows_per_image != rows * C:
            raise ValueError(f'{name}: bad mask')
    return rows, C, ldz, nc


def bn_apply_classify(z, stats, gamma, beta, wp, bp, mask=None, rows_per_image=0):
    """seg[rows, nc] = (mask * Mish(BN(z))) @ wp^T + bp; the activation is never written."""
    dev = _dev(z)
    rows, C, ldz, nc = _cls_args(z, wp, mask, rows_per_image, 'bn_apply_classify')
    _f32(stats, 'stats', 2 * C), _f32(gamma, 'gamma', C), _f32(beta, 'beta', C), _f32(bp, 'bp', nc)
    seg = torch.empty((rows, nc), dtype=z.dtype, device=dev)
    check(lib().ss_bn_apply_classify(dt(z), _p(z), ldz, _p(stats), _p(gamma), _p(beta), _p(mask), rows_per_image, _p(wp), _p(bp), nc,
                                     _p(seg), nc, rows, C, _stream()))
    return seg


def bn_bwd_classify(dseg, z, stats, gamma, beta, wp, mask=None, rows_per_image=0, training=True):
    """-> dz [rows, C], dgb [2,C] (dgamma, dbeta), dwp [nc, C] fp32."""
    dev = _dev(z)
    rows, C, ldz, nc = _cls_args(z, wp, mask, rows_per_image, 'bn_bwd_classify')
    rd, ncd, ldd = _mat(dseg, 'dseg')
    if (rd, ncd) != (rows, nc) or dseg.dtype != z.dtype:
        raise ValueError('bn_bwd_classify: dseg mismatch')
    _f32(stats, 'stats', 2 * C), _f32(gamma, 'gamma', C), _f32(beta, 'beta', C)
    dz = torch.empty((rows, C), dtype=z.dtype, device=dev)
    dgb = torch.empty((2, C), dtype=torch.float32, device=dev)
    dwp = torch.empty((nc, C), dtype=torch.float32, device=dev)
    ws = workspace(dev, (COLRED_BLOCKS * (2 + nc) * C + 2 * C) * 4)
    check(lib().ss_bn_bwd_classify(dt(z), _p(dseg), ldd, _p(z), ldz, _p(stats), _p(gamma), _p(beta), _p(mask), rows_per_image, int(training),
                                   _p(wp), nc, _p(dz), C, _p(dgb), _p(dwp), rows, C, _p(ws), ws.numel(), _stream()))
    return dz, dgb, dwp


# ------------------------------------------------------------------ casts / packing / elementwise
def cast(x, dtype):
    _dev(x)
    if not x.is_contiguous():
        raise ValueError('cast: x must be contiguous')
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    if x.numel():
        check(lib().ss_cast(dt(x), _DT[dtype], _p(x), _p(out), x.numel(), _stream()))
    return out


def transpose_cast(w, dtype, reverse_cols=False):
    """fp32 w [R, Cc] -> [Cc, R] of dtype (optionally reading the columns in reverse order)."""
    _dev(w)
    R, Cc = w.shape
    _f32(w, 'w')
    out = torch.empty((Cc, R), dtype=dtype, device=w.device)
    check(lib().ss_transpose_cast(_DT[dtype], _p(w), _p(out), R, Cc, int(reverse_cols), _stream()))
    return out


def pack_conv_weight(w, dtype, transposed=True, k_multiple=8):
    """fp32 conv weight [O, I, k, k] -> packed [O, Kp] (ky,kx,i order, zero padded) and its transpose [Kp, O];
    Kp = k*k*I rounded up to k_multiple (a multiple of 8)."""
    _dev(w)
    O, I, k, k2 = w.shape
    _f32(w, 'w')
    if k_multiple <= 0 or k_multiple % 8:
        raise ValueError('pack_conv_weight: k_multiple must be a positive multiple of 8')
    Kp = (k * k * I + k_multiple - 1) // k_multiple * k_multiple
    wp = torch.empty((O, Kp), dtype=dtype, device=w.device)
    wpt = torch.empty((Kp, O), dtype=dtype, device=w.device) if transposed else None
    check(lib().ss_pack_conv_weight(_DT[dtype], _p(w), _p(wp), _p(wpt), O, I, k, Kp, _stream()))
    return wp, wpt


def unpack_conv_wgrad(gp, O, I, k):
    _dev(gp)
    Kp = (k * k * I + 7) // 8 * 8
    _f32(gp, 'gp', O * Kp)
    g = torch.empty((O, I, k, k), dtype=torch.float32, device=gp.device)
    check(lib().ss_unpack_conv_wgrad(_p(gp), _p(g), O, I, k, Kp, _stream()))
    return g


def nchw_to_nhwc(x, dtype):
    """fp32 image [B, C, H, W] -> [B*H*W, C] of dtype."""
    _dev(x)
    _f32(x, 'x')
    B, C, H, W = x.shape
    y = torch.empty((B * H * W, C), dtype=dtype, device=x.device)
    check(lib().ss_nchw_to_nhwc(_DT[dtype], _p(x), _p(y), B, C, H * W, _stream()))
    return y


def add(a, b, out=None):
    _dev(a)
    if a.shape != b.shape or a.dtype != b.dtype or not a.is_contiguous() or not b.is_contiguous():
        raise ValueError('add: operands must be alike and contiguous')
    if out is None:
        out = torch.empty_like(a)
    check(lib().ss_add(dt(a), _p(a), _p(b), _p(out), a.numel(), _stream()))
    return out


def rowscale(x, scale, rows_per_scale):
    _dev(x)
    rows, C, ld = _mat(x, 'x')
    if ld != C:
        raise ValueError('rowscale: x must be dense')
    _f32(scale, 'scale')
    if scale.numel() * rows_per_scale < rows:
        raise ValueError('rowscale: scale too short')
    out = torch.empty_like(x)
    check(lib().ss_rowscale(dt(x), _p(x), _p(scale), rows_per_scale, _p(out), rows, C, _stream()))
    return out


def bcast_rows(v, out, rows_per_image):
    """out[b*rows_per_image + r, :] = v[b, :]."""
    _dev(v)
    Bv, C, ldv = _mat(v, 'v')
    rows, Co, ldy = _mat(out, 'out')
    if ldv != C or Co != C or rows != Bv * rows_per_image or v.dtype != out.dtype:
        raise ValueError('bcast_rows: bad shapes')
    check(lib().ss_bcast_rows(dt(v), _p(v), _p(out), ldy, rows_per_image, rows, C, _stream()))
    return out


# ------------------------------------------------------------------ Lawin
def lawin_grid(H, W, r):
    st = 8 // r
    return (H // 8 - 1) * st + 8, (W // 8 - 1) * st + 8


def lawin_pool_fwd(feat, B, H, W, r):
    """feat [B*H*W, C] (row stride ld) -> G [B*gh*gw, C], amax uint8 [B*gh*gw, C]."""
    dev = _dev(feat)
    rows, C, ld = _mat(feat, 'feat')
    if rows != B * H * W:
        raise ValueError('lawin_pool_fwd: bad feat')
    gh, gw = lawin_grid(H, W, r)
    G = torch.empty((B * gh * gw, C), dtype=feat.dtype, device=dev)
    amax = torch.empty((B * gh * gw, C), dtype=torch.uint8, device=dev)
    check(lib().ss_lawin_pool_fwd(dt(feat), _p(feat), ld, _p(G), _p(amax), B, H, W, C, r, _stream()))
    return G, amax


def lawin_pool_bwd(dGs, amaxs, B, H, W, C, dtype, base=None, bvec=None):
    """dGs/amaxs: lists for r = 8, 4, 2 -> d(feat) [B*H*W, C] (= base + pooled-context gradient)."""
    dev = _dev(dGs[0])
    for r, d, a in zip((8, 4, 2), dGs, amaxs):
        gh, gw = lawin_grid(H, W, r)
        _f32(d, 'dG', B * gh * gw * C)
        if a.dtype != torch.uint8 or a.numel() != B * gh * gw * C or not a.is_contiguous():
            raise ValueError('lawin_pool_bwd: bad amax')
    ldb = 0
    if base is not None:
        rb, Cb, ldb = _mat(base, 'base')
        if (rb, Cb) != (B * H * W, C) or base.dtype != dtype:
            raise ValueError('lawin_pool_bwd: bad base')
    _f32(bvec, 'bvec', B * C)
    out = torch.empty((B * H * W, C), dtype=dtype, device=dev)
    check(lib().ss_lawin_pool_bwd(_DT[dtype], _p(dGs[0]), _p(dGs[1]), _p(dGs[2]), _p(amaxs[0]), _p(amaxs[1]), _p(amaxs[2]),
                                  _p(base), ldb, _p(bvec), _p(out), C, B, H, W, C, _stream()))
    return out


def patch_pool_fwd(x, B, H, W, ps):
    """x [B*H*W, C] (row stride ld) -> y [B*ceil(H/ps)*ceil(W/ps), C] = 0.5*(max + avg) pooling with zero right/bottom padding, amax uint8."""
    dev = _dev(x)
    rows, C, ld = _mat(x, 'x')
    if rows != B * H * W:
        raise ValueError('patch_pool_fwd: bad x')
    n = B * (-(-H // ps)) * (-(-W // ps))
    y = torch.empty((n, C), dtype=x.dtype, device=dev)
    amax = torch.empty((n, C), dtype=torch.uint8, device=dev)
    check(lib().ss_patch_pool_fwd(dt(x), _p(x), ld, _p(y), _p(amax), B, H, W, C, ps, _stream()))
    return y, amax


def patch_pool_bwd(dy, amax, B, H, W, ps):
    dev = _dev(dy)
    rows, C, ld = _mat(dy, 'dy')
    n = B * (-(-H // ps)) * (-(-W // ps))
    if rows != n or ld != C or amax.dtype != torch.uint8 or amax.shape != (n, C) or not amax.is_contiguous():
        raise ValueError('patch_pool_bwd: bad dy / amax')
    dx = torch.empty((B * H * W, C), dtype=dy.dtype, device=dev)
    check(lib().ss_patch_pool_bwd(dt(dy), _p(dy), _p(amax), _p(dx), B, H, W, C, ps, _stream()))
    return dx


def lawin_gather_bwd(dctx, B, H, W, r):
    dev = _dev(dctx)
    nwin = B * (H // 8) * (W // 8)
    rows, C, ld = _mat(dctx, 'dctx')
    if rows != nwin * 64 or ld != C:
        raise ValueError('lawin_gather_bwd: bad dctx')
    gh, gw = lawin_grid(H, W, r)
    dG = torch.empty((B * gh * gw, C), dtype=torch.float32, device=dev)
    check(lib().ss_lawin_gather_bwd(dt(dctx), _p(dctx), _p(dG), B, H, W, C, r, _stream()))
    return dG


def lawin_mix(Wm, bm, B, H, W, r, heads, G=None, xwin=None, transpose=False):
    """position mixing; exactly one of G (pooled grid) / xwin (window-ordered [nwin*64, C])."""
    src = G if G is not None else xwin
    dev = _dev(src)
    C = src.shape[1]
    nwin = B * (H // 8) * (W // 8)
    gh, gw = lawin_grid(H, W, r)
    if not src.is_contiguous() or src.shape[0] != (B * gh * gw if G is not None else nwin * 64):
        raise ValueError('lawin_mix: bad source tensor')
    _f32(Wm, 'Wm', heads * 4096)
    _f32(bm, 'bm', heads * 64)
    out = torch.empty((nwin * 64, C), dtype=src.dtype, device=dev)
    ws = workspace(dev, lib().ss_lawin_mix_workspace(heads))
    check(lib().ss_lawin_mix(dt(src), _p(G), _p(xwin), _p(Wm), _p(bm), _p(out), B, H, W, C, r, heads, int(transpose), _p(ws), ws.numel(),
                             _stream()))
    return out


def lawin_mix_wgrad(G, dwin, B, H, W, r, heads):
    """-> dWb [heads, 64, 65] fp32."""
    dev = _dev(G)
    C = G.shape[1]
    nwin = B * (H // 8) * (W // 8)
    if not G.is_contiguous() or not dwin.is_contiguous() or dwin.shape != (nwin * 64, C) or dwin.dtype != G.dtype:
        raise ValueError('lawin_mix_wgrad: bad tensors')
    out = torch.empty((heads, 64, 65), dtype=torch.float32, device=dev)
    ws = workspace(dev, lib().ss_lawin_mix_wgrad_workspace(B, H, W, C, heads))
    check(lib().ss_lawin_mix_wgrad(dt(G), _p(G), _p(dwin), _p(out), B, H, W, C, r, heads, _p(ws), ws.numel(), _stream()))
    return out


def lawin_attn_fwd(theta, pg, B, H, W, heads):
    """theta [B*H*W, inter] (image order, row stride ld), pg [nwin*64, 2*inter] -> y [B*H*W, inter], lse [nwin, heads, 64]."""
    dev = _dev(theta)
    rows, inter, ldt = _mat(theta, 'theta')
    nwin = B * (H // 8) * (W // 8)
    if rows != B * H * W or pg.shape != (nwin * 64, 2 * inter) or not pg.is_contiguous() or pg.dtype != theta.dtype:
        raise ValueError('lawin_attn_fwd: bad shapes')
    y = torch.empty((rows, inter), dtype=theta.dtype, device=dev)
    lse = torch.empty((nwin, heads, 64), dtype=torch.float32, device=dev)
    check(lib().ss_lawin_attn_fwd(dt(theta), _p(theta), ldt, _p(pg), _p(y), inter, _p(lse), B, H, W, inter, heads, _stream()))
    return y, lse


def lawin_attn_bwd(theta, pg, y, dy, lse, B, H, W, heads):
    dev = _dev(theta)
    rows, inter, ldt = _mat(theta, 'theta')
    _, _, ldy = _mat(y, 'y')
    _, _, lddy = _mat(dy, 'dy')
    if y.shape != (rows, inter) or dy.shape != (rows, inter) or dy.dtype != theta.dtype:
        raise ValueError('lawin_attn_bwd: bad shapes')
    dtheta = torch.empty((rows, inter), dtype=theta.dtype, device=dev)
    dpg = torch.empty_like(pg)
    check(lib().ss_lawin_attn_bwd(dt(theta), _p(theta), ldt, _p(pg), _p(y), ldy, _p(dy), lddy, _p(lse), _p(dtheta), inter, _p(dpg),
                                  B, H, W, inter, heads, _stream()))
    return dtheta, dpg


# ------------------------------------------------------------------ losses / metrics
def _nchw(t, name):
    if t.dim() != 4:
        raise ValueError(f'{name}: expected [B, C, H, W]')
    return t.shape, t.stride()


def ohem_ce_fwd(logits, labels, ignore_index, min_kept, thr_state):
    """-> loss (device scalar), prob [B*H*W] fp32, ctl (256-byte device record).  thr_state is updated in place."""
    dev = _dev(logits)
    (B, C, H, W), (sn, sc, sh, sw) = _nchw(logits, 'logits')
    if labels.shape != (B, H, W) or labels.dtype != torch.int64 or not labels.is_contiguous():
        raise ValueError('ohem_ce_fwd: labels must be contiguous int64 [B, H, W]')
    _f32(thr_state, 'thr_state', 1)
    npix = B * H * W
    prob = torch.empty(npix, dtype=torch.float32, device=dev)
    ctl = torch.empty(256, dtype=torch.uint8, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    ws = workspace(dev, lib().ss_ohem_ce_workspace(npix))
    check(lib().ss_ohem_ce_fwd(dt(logits), _p(logits), sn, sc, sh, sw, _p(labels), B, C, H, W, int(ignore_index), float(min_kept),
                               _p(thr_state), _p(prob), _p(ctl), _p(loss), _p(ws), ws.numel(), _stream()))
    return loss, prob, ctl


def ohem_ce_bwd(logits, labels, prob, ctl, gout, ignore_index):
    _dev(logits)
    (B, C, H, W), (sn, sc, sh, sw) = _nchw(logits, 'logits')
    dl = torch.empty_strided(logits.shape, logits.stride(), dtype=logits.dtype, device=logits.device)
    on, oc, oh, ow = dl.stride()
    g = gout.reshape(1).float().contiguous()
    check(lib().ss_ohem_ce_bwd(dt(logits), _p(logits), sn, sc, sh, sw, _p(labels), _p(prob), _p(ctl), _p(g), _p(dl), on, oc, oh, ow,
                               B, C, H, W, int(ignore_index), _stream()))
    return dl


def dice_sums(probs, labels):
    """-> sums [B, C, 3] fp32: (sum p*[y=c], sum p, #[y=c])."""
    dev = _dev(probs)
    (B, C, H, W), (sn, sc, sh, sw) = _nchw(probs, 'probs')
    if labels.shape != (B, H, W) or labels.dtype != torch.int64 or not labels.is_contiguous():
        raise ValueError('dice_sums: labels must be contiguous int64 [B, H, W]')
    sums = torch.empty((B, C, 3), dtype=torch.float32, device=dev)
    ws = workspace(dev, B * 64 * 3 * C * 4)
    check(lib().ss_dice_sums(dt(probs), _p(probs), sn, sc, sh, sw, _p(labels), B, C, H, W, _p(sums), _p(ws), ws.numel(), _stream()))
    return sums


def dice_bwd(probs_like, labels, coef_hit, coef_all, gout):
    _dev(probs_like)
    B, C, H, W = probs_like.shape
    _f32(coef_hit, 'coef_hit', B * C), _f32(coef_all, 'coef_all', B * C)
    d = torch.empty_strided(probs_like.shape, probs_like.stride(), dtype=probs_like.dtype, device=probs_like.device)
    on, oc, oh, ow = d.stride()
    g = gout.reshape(1).float().contiguous()
    check(lib().ss_dice_bwd(dt(probs_like), _p(labels), _p(coef_hit), _p(coef_all), _p(g), _p(d), on, oc, oh, ow, B, C, H, W, _stream()))
    return d


def label_hist(labels, nbins):
    """int64 labels (any shape) -> int64 [nbins] counts of the values 0 .. nbins-1."""
    dev = _dev(labels)
    if labels.dtype != torch.int64 or not labels.is_contiguous():
        raise ValueError('label_hist: labels must be a contiguous int64 tensor')
    if not 0 < nbins <= 4096:
        raise ValueError('label_hist: 1 <= nbins <= 4096')
    hist = torch.zeros(nbins, dtype=torch.int64, device=dev)
    if labels.numel():
        check(lib().ss_label_hist(_p(labels), labels.numel(), nbins, _p(hist), _stream()))
    return hist


def confusion_update(logits, labels, cm):
    """cm int64 [C, C] += counts of (target, argmax(logits, 1))."""
    _dev(logits)
    (B, C, H, W), (sn, sc, sh, sw) = _nchw(logits, 'logits')
    if labels.shape != (B, H, W) or labels.dtype != torch.int64 or not labels.is_contiguous():
        raise ValueError('confusion_update: labels must be contiguous int64 [B, H, W]')
    if cm.dtype != torch.int64 or cm.shape != (C, C) or not cm.is_contiguous():
        raise ValueError('confusion_update: cm must be contiguous int64 [C, C]')
    check(lib().ss_confusion_update(dt(logits), _p(logits), sn, sc, sh, sw, _p(labels), _p(cm), B, C, H, W, _stream()))
    return cm


def confusion_indices(pred, labels, cm):
    """cm int64 [C, C] += counts of (target, pred) for int64 class-index predictions of the labels' shape."""
    _dev(pred)
    if pred.dtype != torch.int64 or labels.dtype != torch.int64 or pred.shape != labels.shape or labels.device != pred.device \
            or not pred.is_contiguous() or not labels.is_contiguous():
        raise ValueError('confusion_indices: pred and labels must be contiguous int64 tensors of one shape on one device')
    if cm.dtype != torch.int64 or cm.dim() != 2 or cm.shape[0] != cm.shape[1] or not cm.is_contiguous():
        raise ValueError('confusion_indices: cm must be contiguous int64 [C, C]')
    if pred.numel():
        check(lib().ss_confusion_indices(_p(pred), _p(labels), _p(cm), pred.numel(), cm.shape[0], _stream()))
    return cm


def _like_channels_last(x):
    B, C, H, W = x.shape
    return torch.empty((B, H, W, C), dtype=x.dtype, device=x.device).permute(0, 3, 1, 2)


def softmax2d_fwd(x):
    """[B, C, H, W] logits (any dense layout) -> probabilities over C, stored channels-last (returned as its NCHW view)."""
    _dev(x)
    (B, C, H, W), sx = _nchw(x, 'x')
    y = _like_channels_last(x)
    check(lib().ss_softmax2d_fwd(dt(x), _p(x), *sx, _p(y), *y.stride(), B, C, H, W, _stream()))
    return y


def softmax2d_bwd(y, dy):
    _dev(y)
    (B, C, H, W), sy = _nchw(y, 'y')
    if dy.shape != y.shape or dy.dtype != y.dtype:
        raise ValueError('softmax2d_bwd: dy must match y')
    dx = _like_channels_last(y)
    check(lib().ss_softmax2d_bwd(dt(y), _p(y), *sy, _p(dy), *dy.stride(), _p(dx), *dx.stride(), B, C, H, W, _stream()))
    return dx


def class_map(prediction):
    """[B, C, H, W] logits (any strides) -> uint8 [B, H, W]: prediction.softmax(1).argmax(1) of helper/callbacks.py:52-54."""
    dev = _dev(prediction)
    (B, C, H, W), (sn, sc, sh, sw) = _nchw(prediction, 'prediction')
    if C > 256:
        raise ValueError('class_map: at most 256 classes fit a uint8 class map')
    out = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    check(lib().ss_class_map_u8(dt(prediction), _p(prediction), sn, sc, sh, sw, _p(out), B, C, H, W, _stream()))
    return out


def bilinear_class_map(x, B, hi, wi, ho, wo, C):
    """x [B*hi*wi, >= C] channels-last logits -> uint8 [B, ho, wo] = class_map(bilinear_fwd(x)[..., :C]) without the resized logits."""
    dev = _dev(x)
    rows, cols, ld = _mat(x, 'x')
    if rows != B * hi * wi or C > cols or C > 256:
        raise ValueError(f'bilinear_class_map: x is [{rows}, {cols}], expected [{B * hi * wi}, >= {C}] with C <= 256')
    out = torch.empty((B, ho, wo), dtype=torch.uint8, device=dev)
    check(lib().ss_bilinear_class_map_u8(dt(x), _p(x), ld, _p(out), B, hi, wi, ho, wo, C, _stream()))
    return out


def confusion_labels(truth, pred, lut, C, valid=None, cm=None):
    """uint8 class maps of one shape -> int64 [C, C] += counts over (lut[truth], lut[pred]) where both are >= 0 and valid != 0."""
    dev = _dev(truth)
    for name, t in (('truth', truth), ('pred', pred)) + ((('valid', valid),) if valid is not None else ()):
        if t.dtype != torch.uint8 or not t.is_contiguous() or t.shape != truth.shape or t.device != dev:
            raise ValueError(f'confusion_labels: {name} must be a contiguous uint8 tensor shaped like truth, on its device')
    if lut.dtype != torch.int32 or lut.numel() != 256 or not lut.is_contiguous() or lut.device != dev:
        raise ValueError('confusion_labels: lut must be a contiguous int32 [256] tensor on the device')
    if cm is None:
        cm = torch.zeros((C, C), dtype=torch.int64, device=dev)
    elif cm.dtype != torch.int64 or cm.shape != (C, C) or not cm.is_contiguous():
        raise ValueError('confusion_labels: cm must be contiguous int64 [C, C]')
    if truth.numel():
        check(lib().ss_confusion_labels_u8(_p(truth), _p(pred), _p(valid) if valid is not None else None, _p(lut), _p(cm), truth.numel(), C,
                                           _stream()))
    return cm


def stage_image(src, mu, sigma, dtype, channels_last):
    """uint8 [B, C, H, W] -> (src - mu[c]) / sigma[c] as a [B, C, H, W] tensor of `dtype`: dense NCHW, or the permuted view of a
    dense [B, H, W, C] map (what runtime.as_tokens_nhwc takes without a copy)."""
    dev = _dev(src)
    if src.dtype != torch.uint8 or src.dim() != 4 or not src.is_contiguous():
        raise ValueError('stage_image: src must be a contiguous uint8 [B, C, H, W] tensor')
    B, C, H, W = src.shape
    for name, t in (('mu', mu), ('sigma', sigma)):
        if t.dtype != torch.float32 or t.shape != (C,) or not t.is_contiguous() or t.device != dev:
            raise ValueError(f'stage_image: {name} must be a contiguous float32 [{C}] tensor on the device')
    if channels_last:
        out = torch.empty((B, H, W, C), dtype=dtype, device=dev).permute(0, 3, 1, 2)
    else:
        out = torch.empty((B, C, H, W), dtype=dtype, device=dev)
    on, oc, oh, ow = out.stride()
    check(lib().ss_stage_image_u8(_DT[dtype], _p(src), _p(mu), _p(sigma), _p(out), on, oc, oh, ow, B, C, H, W, _stream()))
    return out


def _planes(x, who):
    if not x.is_cuda:
        raise RuntimeError('semseg_amd: tensors must live on a HIP device (no CPU fallback)')
    if x.dim() < 2 or not x.is_contiguous() or x.element_size() not in (1, 2, 4, 8):
        raise ValueError(f'{who}: need a contiguous [..., H, W] tensor of 1/2/4/8-byte elements')
    H, W = x.shape[-2:]
    P = x.numel() // (H * W) if H * W else 0
    if P == 0 or P > 65535 or H > 65535:
        raise ValueError(f'{who}: empty tensor, or more than 65535 planes / rows')
    return P, H, W


def flip(x, horizontal):
    """[..., H, W] -> mirrored along W (horizontal) or H; any 1/2/4/8-byte dtype, bit-exact."""
    P, H, W = _planes(x, 'flip')
    out = torch.empty_like(x)
    check(lib().ss_flip(x.element_size(), _p(x), _p(out), P, H, W, 1 if horizontal else 0, _stream()))
    return out


def warp_nearest(x, coef, kind, fill_bits=()):
    """x [..., H, W]; coef float32 [n, 8] on the device (n divides the plane count: plane p uses row p // (planes / n));
    kind 0 affine (inverse matrix m0..m5), 1 perspective (8 coefficients); fill_bits: up to 16 integer bit patterns, plane p takes
    fill_bits[p % len] where its source falls outside the image."""
    import ctypes
    P, H, W = _planes(x, 'warp_nearest')
    if coef.dtype != torch.float32 or coef.dim() != 2 or coef.shape[1] != 8 or not coef.is_contiguous() or coef.device != x.device:
        raise ValueError('warp_nearest: coef must be a contiguous float32 [n, 8] tensor on the device of x')
    n = coef.shape[0]
    if n == 0 or P % n:
        raise ValueError('warp_nearest: the transform count must divide the plane count')
    if len(fill_bits) > 16:
        raise ValueError('warp_nearest: at most 16 fill patterns')
    arr = (ctypes.c_uint64 * max(1, len(fill_bits)))(*[int(v) & 0xFFFFFFFFFFFFFFFF for v in fill_bits])
    out = torch.empty_like(x)
    check(lib().ss_warp_nearest(kind, x.element_size(), _p(x), _p(out), P, H, W, _p(coef), n, arr, len(fill_bits), _stream()))
    return out


def stage_label(src):
    """uint8 labels (any shape) -> int64."""
    dev = _dev(src)
    if src.dtype != torch.uint8 or not src.is_contiguous():
        raise ValueError('stage_label: src must be a contiguous uint8 tensor')
    out = torch.empty(src.shape, dtype=torch.int64, device=dev)
    if src.numel():
        check(lib().ss_stage_label_u8(_p(src), _p(out), src.numel(), _stream()))
    return out


def colsum_any(x):
    """[rows, C] -> [C] fp32 column sums for any C (narrow tensors use the scalar kernel)."""
    rows, C, ld = _mat(x, 'x')
    if C % 8 == 0 and ld % 8 == 0:
        return colsum(x)[0]
    dev = _dev(x)
    out = torch.empty(C, dtype=torch.float32, device=dev)
    ws = workspace(dev, 512 * C * 4)
    check(lib().ss_colsum_small(dt(x), _p(x), ld, rows, C, _p(out), _p(ws), ws.numel(), _stream()))
    return out


def sum_slices(x, nslices, slice_stride, C, bvec=None, rows_per_image=0):
    """out[r, :C] = sum_k x[r, k*slice_stride : k*slice_stride + C] + bvec[r // rows_per_image]."""
    dev = _dev(x)
    rows, _, ld = _mat(x, 'x')
    if bvec is not None:
        _f32(bvec, 'bvec')
    out = torch.empty((rows, C), dtype=x.dtype, device=dev)
    check(lib().ss_sum_slices(dt(x), _p(x), ld, nslices, slice_stride, _p(bvec), rows_per_image, _p(out), rows, C, _stream()))
    return out


# ------------------------------------------------------------------ general depthwise conv / GELU (ConvNeXt, TopFormer)
def dwconvk_fwd(x, wkkc, bias, B, H, W, k, stride=1, pad_top=None, pad_left=None, Ho=None, Wo=None, act=0, res=None):
    """x [B*H*W, C] -> y [B*Ho*Wo, C] (+ res, shaped like y);  wkkc: tap-major [k*k, C] fp32.  Default: symmetric padding k//2, stride 1."""
    dev = _dev(x)
    rows, C, ld = _mat(x, 'x')
    if ld != C or rows != B * H * W:
        raise ValueError('dwconvk_fwd: x must be dense [B*H*W, C]')
    pad_top = k // 2 if pad_top is None else pad_top
    pad_left = k // 2 if pad_left is None else pad_left
    Ho = (H + 2 * pad_top - k) // stride + 1 if Ho is None else Ho
    Wo = (W + 2 * pad_left - k) // stride + 1 if Wo is None else Wo
    _f32(wkkc, 'wkkc', k * k * C), _f32(bias, 'bias', C)
    y = torch.empty((B * Ho * Wo, C), dtype=x.dtype, device=dev)
    if res is not None:
        if res.shape != y.shape or res.dtype != y.dtype or not res.is_contiguous():
            raise ValueError('dwconvk_fwd: res must be a dense tensor shaped like the output')
        check(lib().ss_dwconvk_fwd_res(dt(x), _p(x), _p(wkkc), _p(bias), _p(res), _p(y), B, H, W, C, k, stride, pad_top, pad_left, Ho, Wo, act,
                                       _stream()))
        return y
    check(lib().ss_dwconvk_fwd(dt(x), _p(x), _p(wkkc), _p(bias), _p(y), B, H, W, C, k, stride, pad_top, pad_left, Ho, Wo, act, _stream()))
    return y


def dwconvk_wgrad(g, x, B, H, W, k):
    """-> dw_db [(k*k + 1), C] fp32 (taps tap-major, then the bias gradient); stride 1, padding k//2."""
    dev = _dev(x)
    rows, C, ld = _mat(x, 'x')
    if ld != C or rows != B * H * W or g.shape != x.shape or g.dtype != x.dtype or not g.is_contiguous():
        raise ValueError('dwconvk_wgrad: bad x / g')
    out = torch.empty((k * k + 1, C), dtype=torch.float32, device=dev)
    ws = workspace(dev, COLRED_BLOCKS * (k * k + 1) * C * 4)
    check(lib().ss_dwconvk_wgrad(dt(x), _p(g), _p(x), _p(out), B, H, W, C, k, k // 2, _p(ws), ws.numel(), _stream()))
    return out


def rowdot_f32(a, b):
    _dev(a)
    rows, C = a.shape
    _f32(a, 'a'), _f32(b, 'b', rows * C)
    out = torch.empty(rows, dtype=torch.float32, device=a.device)
    check(lib().ss_rowdot_f32(_p(a), _p(b), _p(out), rows, C, _stream()))
    return out


def gemm_nt_gelu2(a, w, bias=None):
    """z = a @ w.T + bias and GELU(z), both kept: -> (z, gelu) in a's dtype (one GEMM epilogue on the bf16 tile kernels)."""
    dev = _dev(a)
    M, K, lda = _mat(a, 'a')
    N, K2, ldb = _mat(w, 'w')
    if K2 != K or w.dtype != a.dtype:
        raise ValueError(f'gemm_nt_gelu2: a is [{M}, {K}] {a.dtype}, w is [{N}, {K2}] {w.dtype}')
    _f32(bias, 'bias', N)
    z = torch.empty((M, N), dtype=a.dtype, device=dev)
    g = torch.empty((M, N), dtype=a.dtype, device=dev)
    check(lib().ss_gemm_nt_gelu2(dt(a), _p(a), lda, _p(w), ldb, _p(z), N, _p(g), N, M, N, K, _p(bias), _stream()))
    return z, g


def gelu_fwd(z):
    _dev(z)
    if not z.is_contiguous():
        raise ValueError('gelu_fwd: z must be contiguous')
    a = torch.empty_like(z)
    check(lib().ss_gelu_fwd(dt(z), _p(z), _p(a), z.numel(), _stream()))
    return a


def gelu_bwd(da, z):
    _dev(z)
    if not z.is_contiguous() or not da.is_contiguous() or da.shape != z.shape or da.dtype != z.dtype:
        raise ValueError('gelu_bwd: da / z must be contiguous and alike')
    dz = torch.empty_like(z)
    check(lib().ss_gelu_bwd(dt(z), _p(da), _p(z), _p(dz), z.numel(), _stream()))
    return dz


# ------------------------------------------------------------------ Decoder / DNet tail
def decoder_stats(z9, B, H4, W4, oc, dbias, eps=1e-5, momentum=0.1, running_mean=None, running_var=None, nbt=None):
    dev = _dev(z9)
    rows, _, ld9 = _mat(z9, 'z9')
    if rows != B * H4 * W4:
        raise ValueError('decoder_stats: bad z9')
    _f32(dbias, 'dbias', oc)
    stats = torch.empty(2 * oc, dtype=torch.float32, device=dev)
    ws = workspace(dev, COLRED_BLOCKS * 2 * ld9 * 4)
    check(lib().ss_decoder_stats(dt(z9), _p(z9), ld9, B, H4, W4, oc, _p(dbias), eps, momentum, _p(stats), _p(running_mean), _p(running_var),
                                 _p(nbt), _p(ws), ws.numel(), _stream()))
    return stats


def decoder_combine_fwd(z9, bs, dbias, stats, gamma, beta, B, H4, W4, oc):
    dev = _dev(z9)
    _, _, ld9 = _mat(z9, 'z9')
    _, _, ldb = _mat(bs, 'bs')
    prod = torch.empty((B * (4 * H4 - 1) * (4 * W4 - 1), oc), dtype=z9.dtype, device=dev)
    check(lib().ss_decoder_combine_fwd(dt(z9), _p(z9), ld9, _p(bs), ldb, _p(dbias), _p(stats), _p(gamma), _p(beta), _p(prod), B, H4, W4, oc, _stream()))
    return prod


def decoder_combine_bwd(dprod, z9, bs, dbias, stats, gamma, beta, training, B, H4, W4, oc):
    """-> dz9 (like z9), dbs (like bs), dgb [2*oc] (dgamma, dbeta), d(deconv bias) [oc]."""
    dev = _dev(z9)
    _, _, ld9 = _mat(z9, 'z9')
    _, _, ldb = _mat(bs, 'bs')
    if not dprod.is_contiguous() or dprod.dtype != z9.dtype or dprod.shape != (B * (4 * H4 - 1) * (4 * W4 - 1), oc) or not z9.is_contiguous() or not bs.is_contiguous():
        raise ValueError('decoder_combine_bwd: bad tensors')
    dz9, dbs = torch.empty_like(z9), torch.empty_like(bs)
    dgb = torch.empty(2 * oc, dtype=torch.float32, device=dev)
    dbg = torch.empty(oc, dtype=torch.float32, device=dev)
    ws = workspace(dev, (COLRED_BLOCKS * 2 * oc + 4 * oc) * 4)
    check(lib().ss_decoder_combine_bwd(dt(z9), _p(dprod), _p(z9), ld9, _p(bs), ldb, _p(dbias), _p(stats), _p(gamma), _p(beta), int(training),
                                       _p(dz9), _p(dbs), _p(dgb), _p(dbg), B, H4, W4, oc, _p(ws), ws.numel(), _stream()))
    return dz9, dbs, dgb, dbg


def terminator_fwd(prod, wt, bt, B, Hd, Wd, H, W, probs_f32=False):
    """prod [B*Hd*Wd, oc] -> probabilities NCHW [B, nc, H, W] (fp32 when probs_f32, else prod's dtype)."""
    dev = _dev(prod)
    nc, oc = wt.shape
    _f32(wt, 'wt'), _f32(bt, 'bt', nc)
    if not prod.is_contiguous() or prod.shape != (B * Hd * Wd, oc):
        raise ValueError('terminator_fwd: bad prod')
    probs = torch.empty((B, nc, H, W), dtype=torch.float32 if probs_f32 else prod.dtype, device=dev)
    check(lib().ss_terminator_fwd(dt(prod), int(probs_f32), _p(prod), _p(wt), _p(bt), _p(probs), B, Hd, Wd, H, W, oc, nc, _stream()))
    return probs


def terminator_bwd(dprobs, prod, wt, bt, B, Hd, Wd, H, W):
    """-> dprod [B*Hd*Wd, oc], dW [nc, oc] fp32, db [nc] fp32."""
    dev = _dev(prod)
    nc, oc = wt.shape
    if not dprobs.is_contiguous() or dprobs.dtype not in (prod.dtype, torch.float32) or dprobs.shape != (B, nc, H, W):
        raise ValueError('terminator_bwd: dprobs must be contiguous NCHW like the probabilities')
    probs_f32 = dprobs.dtype == torch.float32
    dprod = torch.empty_like(prod)
    out = torch.empty(nc * oc + nc, dtype=torch.float32, device=dev)
    ws = workspace(dev, COLRED_BLOCKS * (nc * oc + nc) * 4)
    check(lib().ss_terminator_bwd(dt(prod), int(probs_f32), _p(dprobs), _p(prod), _p(wt), _p(bt), _p(dprod), _p(out), B, Hd, Wd, H, W, oc, nc, _p(ws), ws.numel(),
                                  _stream()))
    return dprod, out[:nc * oc].view(nc, oc), out[nc * oc:]


# ------------------------------------------------------------------ TopFormer helpers
def ir_block_ok(dtype, Cin, hidp, Cout, k, s):
    return bool(lib().ss_ir_block_ok(1 if dtype == torch.bfloat16 else 0, Cin, hidp, Cout, k, s))


def ir_block_fwd(x, B, H, W, Cout, k, s, pad_top, pad_left, Ho, Wo, w_in, b_in, act_in, w_dw, b_dw, act_dw, w_out, b_out, residual=False):
    """One fused inverted-residual block (ss_ir_block_fwd): x [B*H*W, Cin] -> y [B*Ho*Wo, Cout]; weights padded as the header says."""
    dev = _dev(x)
    rows, Cin, ld = _mat(x, 'x')
    hidp, cinp = w_in.shape
    if rows != B * H * W or ld != Cin or not x.is_contiguous():
        raise ValueError('ir_block_fwd: x must be dense [B*H*W, Cin]')
    if (cinp != (Cin + 15) // 16 * 16 or tuple(w_dw.shape) != (k * k, hidp) or tuple(w_out.shape) != ((Cout + 31) // 32 * 32, hidp)
            or w_in.dtype != x.dtype or w_out.dtype != x.dtype or not (w_in.is_contiguous() and w_out.is_contiguous())):
        raise ValueError('ir_block_fwd: weights are not padded as ss_ir_block_fwd expects')
    _f32(b_in, 'b_in', hidp), _f32(b_dw, 'b_dw', hidp), _f32(w_dw, 'w_dw', k * k * hidp), _f32(b_out, 'b_out', w_out.shape[0])
    y = torch.empty((B * Ho * Wo, Cout), dtype=x.dtype, device=dev)
    check(lib().ss_ir_block_fwd(dt(x), _p(x), _p(y), B, H, W, Cin, hidp, Cout, k, s, pad_top, pad_left, Ho, Wo, _p(w_in), _p(b_in), act_in,
                                _p(w_dw), _p(b_dw), act_dw, _p(w_out), _p(b_out), int(residual), _stream()))
    return y


def im2col_ex(x, B, H, W, C, k, s, pad_top, pad_left, Ho, Wo):
    dev = _dev(x)
    if not x.is_contiguous() or x.numel() != B * H * W * C:
        raise ValueError('im2col_ex: bad x')
    Kp = (k * k * C + 7) // 8 * 8
    col = torch.empty((B * Ho * Wo, Kp), dtype=x.dtype, device=dev)
    check(lib().ss_im2col_ex(dt(x), _p(x), _p(col), B, H, W, C, k, s, pad_top, pad_left, Ho, Wo, Kp, _stream()))
    return col


def stem_conv(x, wp, bias, pad_top, pad_left, Ho, Wo, dtype, act=0):
    """x [B, 3, H, W] (fp32 or `dtype`, any dense layout) -> [B*Ho*Wo, 16] in `dtype`: 3x3 / stride 2 conv with the packed weight
    wp [16, Kp] of pack_conv_weight + bias + activation, no im2col matrix."""
    dev = _dev(x)
    B, C, H, W = x.shape
    if x.dtype not in (torch.float32, dtype) or wp.dtype != dtype or wp.shape[0] != 16 or not wp.is_contiguous():
        raise ValueError('stem_conv: x must be fp32 or the compute dtype, wp a contiguous [16, Kp] matrix in the compute dtype')
    y = torch.empty((B * Ho * Wo, 16), dtype=dtype, device=dev)
    sn, sc, sh, sw = x.stride()
    check(lib().ss_stem_conv3x3s2(_DT[dtype], _DT[x.dtype], _p(x), sn, sc, sh, sw, _p(wp), wp.shape[1], _p(bias) if bias is not None else None,
                                  _p(y), B, C, H, W, pad_top, pad_left, Ho, Wo, act, _stream()))
    return y


def avgpool(x, B, H, W, win, out=None):
    dev = _dev(x)
    rows, C, ldx = _mat(x, 'x')
    if rows != B * H * W:
        raise ValueError('avgpool: bad x')
    ro = B * (H // win) * (W // win)
    if out is None:
        out = torch.empty((ro, C), dtype=x.dtype, device=dev)
    r2, C2, ldy = _mat(out, 'out')
    if (r2, C2) != (ro, C) or out.dtype != x.dtype:
        raise ValueError('avgpool: bad out')
    check(lib().ss_avgpool(dt(x), _p(x), ldx, _p(out), ldy, B, H, W, C, win, _stream()))
    return out


def act(x, code):
    _dev(x)
    if not x.is_contiguous():
        raise ValueError('act: x must be contiguous')
    y = torch.empty_like(x)
    check(lib().ss_act(dt(x), _p(x), _p(y), x.numel(), code, _stream()))
    return y


def inject_fuse(local, gact, gemb, B, hl, wl, hg, wg):
    dev = _dev(local)
    rows, C, ldl = _mat(local, 'local')
    ra, Ca, lda = _mat(gact, 'gact')
    re_, Ce, lde = _mat(gemb, 'gemb')
    if rows != B * hl * wl or ra != B * hg * wg or re_ != ra or Ca != C or Ce != C or gact.dtype != local.dtype or gemb.dtype != local.dtype:
        raise ValueError('inject_fuse: bad shapes')
    out = torch.empty((rows, C), dtype=local.dtype, device=dev)
    check(lib().ss_inject_fuse(dt(local), _p(local), ldl, _p(gact), lda, _p(gemb), lde, _p(out), C, B, hl, wl, hg, wg, C, _stream()))
    return out


def inject_fuse_sum(local, gact, gemb, B, hl, wl, hg, wg, extra):
    """local * up(gact) + up(gemb) + sum of up(t) over extra = [(t [B*h*w, C], h, w), ...] (at most 3), one pass."""
    import ctypes
    dev = _dev(local)
    rows, C, ldl = _mat(local, 'local')
    ra, Ca, lda = _mat(gact, 'gact')
    re_, Ce, lde = _mat(gemb, 'gemb')
    if rows != B * hl * wl or ra != B * hg * wg or re_ != ra or Ca != C or Ce != C or gact.dtype != local.dtype or gemb.dtype != local.dtype:
        raise ValueError('inject_fuse_sum: bad shapes')
    n = len(extra)
    if n > 3:
        raise ValueError('inject_fuse_sum: at most 3 extra sources')
    ptrs, lds, hs, ws = (ctypes.c_void_p * 3)(), (ctypes.c_int64 * 3)(), (ctypes.c_int * 3)(), (ctypes.c_int * 3)()
    for k, (t, h, w) in enumerate(extra):
        r, Ct, ldt = _mat(t, 'extra')
        if r != B * h * w or Ct != C or t.dtype != local.dtype:
            raise ValueError('inject_fuse_sum: bad extra source')
        ptrs[k], lds[k], hs[k], ws[k] = t.data_ptr(), ldt, h, w
    out = torch.empty((rows, C), dtype=local.dtype, device=dev)
    check(lib().ss_inject_fuse_sum(dt(local), _p(local), ldl, _p(gact), lda, _p(gemb), lde, n, ptrs, lds, hs, ws, _p(out), C, B, hl, wl, hg, wg, C,
                                   _stream()))
    return out


def bilinear_add(x, y, B, hi, wi, ho, wo):
    """y [B*ho*wo, C] += bilinear(x [B*hi*wi, C])."""
    _dev(x)
    rows, C, ldx = _mat(x, 'x')
    ro, Co, ldy = _mat(y, 'y')
    if rows != B * hi * wi or ro != B * ho * wo or Co != C or x.dtype != y.dtype:
        raise ValueError('bilinear_add: bad shapes')
    check(lib().ss_bilinear_add(dt(x), _p(x), ldx, _p(y), ldy, B, hi, wi, ho, wo, C, _stream()))
    return y


# ---------------------------------------------------------------------------------- TopFormer RefinerHead (inference)
SIGMOID = 5
GELU_GRAD = 6            # ss_gemm_nt epilogue: times GELU'(res)


def image_moments(x, images, pre_act=0):
    """x [images*n, C] -> moments [images, 2, C] fp32: per-image mean of f(x) and sum (f(x) - mean)^2 (f = sigmoid if pre_act == 5)."""
    dev = _dev(x)
    rows, C, ld = _mat(x, 'x')
    if images <= 0 or rows % images:
        raise ValueError('image_moments: rows not divisible by images')
    out = torch.empty((images, 2, C), dtype=torch.float32, device=dev)
    ws = workspace(dev, images * COLRED_BLOCKS * 2 * C * 4)
    check(lib().ss_image_moments(dt(x), _p(x), ld, rows // images, C, images, pre_act, _p(out), _p(ws), ws.numel(), _stream()))
    return out


def simam_apply(x, moments, k, images, out=None):
    dev = _dev(x)
    rows, C, ldx = _mat(x, 'x')
    _f32(moments, 'moments', images * 2 * C), _f32(k, 'k', 1)
    if rows % images:
        raise ValueError('simam_apply: rows not divisible by images')
    if out is None:
        out = torch.empty((rows, C), dtype=x.dtype, device=dev)
    ro, Co, ldy = _mat(out, 'out')
    if (ro, Co) != (rows, C) or out.dtype != x.dtype:
        raise ValueError('simam_apply: bad out')
    check(lib().ss_simam_apply(dt(x), _p(x), ldx, _p(moments), _p(k), _p(out), ldy, rows // images, C, images, _stream()))
    return out


def simam_gemm(x, moments, k, images, w, out=None):
    """(SimAM(x) rounded to x's dtype) @ w.T -> fp32 [rows, N]; one kernel where ss_simam_gemm takes the shape, else simam_apply + gemm_nt."""
    dev = _dev(x)
    rows, C, ldx = _mat(x, 'x')
    N, Cw, ldw = _mat(w, 'w')
    if Cw != C or w.dtype != x.dtype or rows % images:
        raise ValueError('simam_gemm: bad shapes')
    _f32(moments, 'moments', images * 2 * C), _f32(k, 'k', 1)
    if out is None:
        out = torch.empty((rows, N), dtype=torch.float32, device=dev)
    ro, No, ldo = _mat(out, 'out')
    if (ro, No) != (rows, N) or out.dtype != torch.float32:
        raise ValueError('simam_gemm: bad out')
    if not lib().ss_simam_gemm_ok(dt(x), C, N) or ldo % 4 or out.data_ptr() % 16:
        return gemm_nt(simam_apply(x, moments, k, images), w, out=out, out_f32=True)
    check(lib().ss_simam_gemm(dt(x), _p(x), ldx, _p(moments), _p(k), _p(w), ldw, _p(out), ldo, rows // images, C, images, N, _stream()))
    return out


def refiner_fold(moments, rows_per_image, groups, cg, pitch, eps, gamma, beta, L, mrows):
    """GroupNorm(groups, groups*cg) on columns g*pitch + j of an [.., ncol] map, then the fixed map L [lrows, groups*cg]
    ->  (M [images, mrows, ncol], c [images, mrows])."""
    dev = _dev(moments)
    images, two, ncol = moments.shape
    _f32(moments, 'moments'), _f32(gamma, 'gamma', groups * cg), _f32(beta, 'beta', groups * cg)
    lrows = L.shape[0]
    _f32(L, 'L', lrows * groups * cg)
    M = torch.empty((images, mrows, ncol), dtype=torch.float32, device=dev)
    c = torch.empty((images, mrows), dtype=torch.float32, device=dev)
    check(lib().ss_refiner_fold(_p(moments), rows_per_image, images, groups, cg, pitch, ncol, eps, _p(gamma), _p(beta), _p(L), lrows, mrows, _p(M),
                                _p(c), _stream()))
    return M, c


def image_affine(x, M, c, out_dtype=torch.float32, pre_act=0):
    """x fp32 [images*n, cin] -> out [images*n, mrows] = M[img] f(x) + c[img]."""
    dev = _dev(x)
    rows, cin, ldi = _mat(x, 'x')
    images, mrows, cin2 = M.shape
    if x.dtype != torch.float32 or cin2 != cin or rows % images:
        raise ValueError('image_affine: bad shapes')
    _f32(M, 'M'), _f32(c, 'c', images * mrows)
    out = torch.empty((rows, mrows), dtype=out_dtype, device=dev)
    code = {torch.float32: 0, torch.bfloat16: 1}[out_dtype]
    check(lib().ss_image_affine(code, _p(x), ldi, _p(M), _p(c), _p(out), mrows, rows // images, images, cin, mrows, pre_act, _stream()))
    return out


# ---------------------------------------------------------------------------------- TopFormer training (backward-side kernels)
def dwconvk_dgrad(g, wkkc, B, H, W, k, stride, pad_top, pad_left, Ho, Wo):
    """g [B*Ho*Wo, C] -> dx [B*H*W, C]: data gradient of dwconvk_fwd (wkkc: the forward's tap-major [k*k, C] fp32 weight)."""
    dev = _dev(g)
    rows, C, ld = _mat(g, 'g')
    if ld != C or rows != B * Ho * Wo:
        raise ValueError('dwconvk_dgrad: g must be dense [B*Ho*Wo, C]')
    _f32(wkkc, 'wkkc', k * k * C)
    dx = torch.empty((B * H * W, C), dtype=g.dtype, device=dev)
    check(lib().ss_dwconvk_dgrad(dt(g), _p(g), _p(wkkc), _p(dx), B, H, W, C, k, stride, pad_top, pad_left, Ho, Wo, _stream()))
    return dx


def dwconvk_wgrad_s(g, x, B, H, W, k, stride, pad_top, pad_left, Ho, Wo):
    """-> dw_db [(k*k + 1), C] fp32 (taps tap-major, then the bias gradient) for stride 1 / 2 and asymmetric padding; k in {3, 5}."""
    dev = _dev(x)
    rows, C, ld = _mat(x, 'x')
    rg, Cg, ldg = _mat(g, 'g')
    if ld != C or rows != B * H * W or ldg != C or Cg != C or rg != B * Ho * Wo or g.dtype != x.dtype:
        raise ValueError('dwconvk_wgrad_s: bad x / g')
    out = torch.empty((k * k + 1, C), dtype=torch.float32, device=dev)
    ws = workspace(dev, 512 * (k * k + 1) * C * 4)
    check(lib().ss_dwconvk_wgrad_s(dt(x), _p(g), _p(x), _p(out), B, H, W, C, k, stride, pad_top, pad_left, Ho, Wo, _p(ws), ws.numel(), _stream()))
    return out


def avgpool_bwd(g, B, H, W, win):
    """g [B*(H/win)*(W/win), C] (may be a channel slice) -> dx [B*H*W, C]."""
    dev = _dev(g)
    rows, C, ldg = _mat(g, 'g')
    if rows != B * (H // win) * (W // win):
        raise ValueError('avgpool_bwd: bad g')
    dx = torch.empty((B * H * W, C), dtype=g.dtype, device=dev)
    check(lib().ss_avgpool_bwd(dt(g), _p(g), ldg, _p(dx), C, B, H, W, C, win, _stream()))
    return dx


def act_bwd(dy, y, code):
    """dx = dy * act'(.) from the activation's output y (2 ReLU, 3 ReLU6, 5 sigmoid)."""
    _dev(dy)
    if dy.shape != y.shape or dy.dtype != y.dtype or not dy.is_contiguous() or not y.is_contiguous():
        raise ValueError('act_bwd: dy / y must be alike and contiguous')
    dx = torch.empty_like(dy)
    check(lib().ss_act_bwd(dt(dy), _p(dy), _p(y), _p(dx), dy.numel(), code, _stream()))
    return dx


def mul(a, b):
    _dev(a)
    if a.shape != b.shape or a.dtype != b.dtype or not a.is_contiguous() or not b.is_contiguous():
        raise ValueError('mul: operands must be alike and contiguous')
    out = torch.empty_like(a)
    check(lib().ss_mul(dt(a), _p(a), _p(b), _p(out), a.numel(), _stream()))
    return out


def inject_bwd(g, local, gact, B, hl, wl, hg, wg):
    """InjectionDotSum backward -> (dlocal, dgact, dgemb); one pass over the fine map where ss_inject_bwd takes the geometry, else the resize,
    two products and two transposed resizes."""
    dev = _dev(g)
    rows, C, ldg = _mat(g, 'g')
    ra, Ca, lda = _mat(gact, 'gact')
    if rows != B * hl * wl or ra != B * hg * wg or Ca != C or local.shape != g.shape or local.dtype != g.dtype or gact.dtype != g.dtype:
        raise ValueError('inject_bwd: bad shapes')
    if not (lib().ss_inject_bwd_ok(hl, wl, hg, wg, C) and g.is_contiguous() and local.is_contiguous()):
        dlocal = mul(g, bilinear_fwd(gact, B, hg, wg, hl, wl))
        return dlocal, bilinear_bwd(mul(g, local), B, hg, wg, hl, wl), bilinear_bwd(g, B, hg, wg, hl, wl)
    dlocal = torch.empty_like(g)
    dgact = torch.empty((ra, C), dtype=g.dtype, device=dev)
    dgemb = torch.empty((ra, C), dtype=g.dtype, device=dev)
    ws = workspace(dev, B * (hg + 1) * (wg + 1) * 8 * C * 4)
    check(lib().ss_inject_bwd(dt(g), _p(g), _p(local), _p(gact), lda, _p(dlocal), _p(dgact), C, _p(dgemb), C, B, hl, wl, hg, wg, C, _p(ws), ws.numel(),
                              _stream()))
    return dlocal, dgact, dgemb


def simam_bwd(dy, x, moments, k, images, dk, accumulate_dk=True):
    """-> dx (like x); dk (one fp32) is written or accumulated in place."""
    dev = _dev(x)
    rows, C, ldx = _mat(x, 'x')
    rd, Cd, ldd = _mat(dy, 'dy')
    if (rd, Cd) != (rows, C) or dy.dtype != x.dtype or rows % images:
        raise ValueError('simam_bwd: bad shapes')
    _f32(moments, 'moments', images * 2 * C), _f32(k, 'k', 1), _f32(dk, 'dk', 1)
    dx = torch.empty((rows, C), dtype=x.dtype, device=dev)
    sums = torch.empty((images, 2, C), dtype=torch.float32, device=dev)
    ws = workspace(dev, images * COLRED_BLOCKS * 2 * C * 4)
    check(lib().ss_simam_bwd(dt(x), _p(dy), ldd, _p(x), ldx, _p(moments), _p(k), _p(dx), C, _p(dk), int(accumulate_dk), _p(sums), rows // images, C, images,
                             _p(ws), ws.numel(), _stream()))
    return dx


def groupnorm_fwd(x, images, groups, cg, pitch, eps, gamma, beta, pre_act=0):
    """fp32 x [images*n, ncol] -> (y, stats [images, groups, 2]); column g*pitch + j (j < cg) is channel g*cg + j, the rest padding."""
    dev = _dev(x)
    rows, ncol, ldx = _mat(x, 'x')
    if x.dtype != torch.float32 or rows % images:
        raise ValueError('groupnorm_fwd: x must be fp32 [images*n, ncol]')
    _f32(gamma, 'gamma', groups * cg), _f32(beta, 'beta', groups * cg)
    mom = image_moments(x, images, pre_act=pre_act)
    coef = torch.empty((images, 2, ncol), dtype=torch.float32, device=dev)
    stats = torch.empty((images, groups, 2), dtype=torch.float32, device=dev)
    y = torch.empty((rows, ncol), dtype=torch.float32, device=dev)
    check(lib().ss_groupnorm_fwd(_p(x), ldx, _p(mom), rows // images, images, groups, cg, pitch, ncol, eps, _p(gamma), _p(beta), pre_act, _p(coef), _p(stats),
                                 _p(y), ncol, _stream()))
    return y, stats


def groupnorm_bwd(dy, x, stats, gamma, images, groups, cg, pitch, pre_act=0):
    """-> dx [like x], dgb [2, groups*cg] fp32 (dgamma, dbeta)."""
    dev = _dev(x)
    rows, ncol, ldx = _mat(x, 'x')
    rd, nd, ldd = _mat(dy, 'dy')
    if x.dtype != torch.float32 or dy.dtype != torch.float32 or (rd, nd) != (rows, ncol) or rows % images:
        raise ValueError('groupnorm_bwd: bad shapes')
    _f32(stats, 'stats', images * groups * 2), _f32(gamma, 'gamma', groups * cg)
    dx = torch.empty((rows, ncol), dtype=torch.float32, device=dev)
    dgb = torch.empty((2, groups * cg), dtype=torch.float32, device=dev)
    scratch = torch.empty(images * 5 * ncol, dtype=torch.float32, device=dev)
    ws = workspace(dev, images * COLRED_BLOCKS * 2 * ncol * 4)
    check(lib().ss_groupnorm_bwd(_p(dy), ldd, _p(x), ldx, _p(stats), _p(gamma), rows // images, images, groups, cg, pitch, ncol, pre_act, _p(dx), ncol,
                                 _p(dgb), _p(scratch), _p(ws), ws.numel(), _stream()))
    return dx, dgb


# ---------------------------------------------------------------------------------- gradient delivery (engine.FlatParams)
def multi_add_layout(t):
    """-> (cols, ld) if `t` is a source ss_multi_add_f32 can read (contiguous, a strided 1-D vector, or 2-D rows with unit inner
    stride), else None."""
    if t.is_contiguous():
        return t.numel(), t.numel()
    if t.dim() == 1:
        return 1, t.stride(0)
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]:
        return t.shape[1], t.stride(0)
    return None


def multi_add(srcs, dsts):
    """dsts[i] += srcs[i] for lists of fp32 device tensors (dsts contiguous, srcs as multi_add_layout allows), one launch per 40."""
    import ctypes
    n = len(srcs)
    if n == 0:
        return
    lay = []
    for s, d in zip(srcs, dsts):
        _dev(s)
        la = multi_add_layout(s)
        if s.dtype != torch.float32 or d.dtype != torch.float32 or la is None or not d.is_contiguous() or s.numel() != d.numel() \
                or s.numel() == 0 or s.device != d.device:
            raise ValueError('multi_add: need float32 pairs of equal size on one device, contiguous destinations')
        lay.append(la)
    sp = (ctypes.c_void_p * n)(*[s.data_ptr() for s in srcs])
    dp = (ctypes.c_void_p * n)(*[d.data_ptr() for d in dsts])
    ne = (ctypes.c_int64 * n)(*[s.numel() for s in srcs])
    co = (ctypes.c_int64 * n)(*[c for c, _ in lay])
    ld = (ctypes.c_int64 * n)(*[l for _, l in lay])
    check(lib().ss_multi_add_f32(n, sp, ld, co, dp, ne, _stream()))


def droppath_keep(u, keep_prob):
    """u: torch.rand draw [b, ...] in the activation dtype -> fp32 [b] multipliers floor(kp + u) / kp."""
    _dev(u)
    if not u.is_contiguous():
        raise ValueError('droppath_keep: u must be contiguous')
    out = torch.empty(u.numel(), dtype=torch.float32, device=u.device)
    check(lib().ss_droppath_keep(dt(u), _p(u), float(keep_prob), _p(out), u.numel(), _stream()))
    return out
