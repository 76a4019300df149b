"""Fused Mix-FFN (ss_mixffn_fwd / ss_mixffn_bwd) against the launches it replaces, at the MiT-B2 stage-1 / stage-2 shapes of configs[1]."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd import ops

BF = torch.bfloat16


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, H, W, C) in [(16, 256, 256, 64), (16, 128, 128, 128)]:
    Ch, T = 4 * C, B * H * W
    g = torch.Generator(device='cuda').manual_seed(0)
    r = lambda *s, sc=1.0: torch.randn(*s, device='cuda', generator=g) * sc
    x, res = r(T, C).to(BF), r(T, C).to(BF)
    w1, w2 = r(Ch, C, sc=C ** -0.5).to(BF), r(C, Ch, sc=Ch ** -0.5).to(BF)
    b1, bd, b2, w9c = 0.1 * r(Ch), 0.1 * r(Ch), 0.1 * r(C), r(9, Ch, sc=0.3)
    keep = torch.ones(B, device='cuda')

    def three():
        h1 = ops.gemm_nt(x, w1, bias=b1)
        a2 = ops.dwconv3_fwd(h1, w9c, bd, B, H, W, act=1)
        return ops.gemm_nt(a2, w2, bias=b2, res=res, rowscale=keep, rows_per_scale=H * W)
    t3 = timeit(three)
    tf = timeit(lambda: ops.mixffn_fwd(x, w1, b1, w9c, bd, w2, b2, B, H, W, res=res, keep=keep))
    ts = timeit(lambda: ops.mixffn_fwd(x, w1, b1, w9c, bd, w2, b2, B, H, W, res=res, keep=keep, save_hidden=True))
    print(f'fwd  B={B} {H}x{W} C={C}: three launches {t3:7.1f} us   fused {tf:7.1f} us   fused + h1/a2 stored {ts:7.1f} us')
    if hasattr(ops, 'mixffn_bwd'):
        out, h1, a2 = ops.mixffn_fwd(x, w1, b1, w9c, bd, w2, b2, B, H, W, res=res, keep=keep, save_hidden=True)
        dout = r(T, C).to(BF)
        w1t, w2t = w1.t().contiguous(), w2.t().contiguous()

        def three_bwd():
            dW2, db2 = ops.gemm_tn(dout, a2, colsum=True)
            da2 = ops.gemm_nt(dout, w2t)
            dh1, dwdb = ops.dwconv3_bwd_fused(da2, h1, w9c, bd, B, H, W, act=1)
            dW1, db1 = ops.gemm_tn(dh1, x, colsum=True)
            return ops.gemm_nt(dh1, w1t)
        tb3 = timeit(three_bwd)
        tbf = timeit(lambda: ops.mixffn_bwd(dout, x, w1, b1, w9c, bd, w2, B, H, W, keep=keep))
        print(f'bwd  B={B} {H}x{W} C={C}: five launches {tb3:7.1f} us   fused {tbf:7.1f} us')
