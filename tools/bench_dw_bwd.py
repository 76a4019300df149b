"""Fused depthwise 3x3 + GELU backward at the Mix-FFN shapes of the headline step: LDS-exchange kernel (SS_DW_LANE=0) against the
lane-per-column kernel, same inputs, results compared bit for bit.  python tools/bench_dw_bwd.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semseg_amd import ops  # noqa: E402


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    torch.manual_seed(0)
    for B, H, W, C in [(16, 256, 256, 256), (16, 128, 128, 512), (16, 64, 64, 1280), (16, 32, 32, 2048), (2, 40, 70, 64)]:
        x = torch.randn(B * H * W, C, device='cuda').bfloat16()
        dy = torch.randn(B * H * W, C, device='cuda').bfloat16()
        w9c = (torch.randn(9, C, device='cuda') / 3).contiguous()
        b = 0.1 * torch.randn(C, device='cuda')
        out = {}
        for mode in ('0', '1'):
            os.environ['SS_DW_LANE'] = mode
            dx, dwdb = ops.dwconv3_bwd_fused(dy, x, w9c, b, B, H, W, act=1)
            t = timed(lambda: ops.dwconv3_bwd_fused(dy, x, w9c, b, B, H, W, act=1))
            out[mode] = (dx.clone(), dwdb.clone(), t)
        same = torch.equal(out['0'][0], out['1'][0]) and torch.equal(out['0'][1], out['1'][1])
        print(f'{B}x{H}x{W}x{C}: exchange {out["0"][2]:8.1f} us   lane {out["1"][2]:8.1f} us   identical {same}', flush=True)
    os.environ.pop('SS_DW_LANE', None)




def fc2():
    """GEMM (fc2 data gradient) + fused depthwise backward against the one kernel that forms da2 itself."""
    torch.manual_seed(0)
    for B, H, W, C, K in [(16, 256, 256, 256, 64), (16, 128, 128, 512, 128)]:
        x = torch.randn(B * H * W, C, device='cuda').bfloat16()
        dout = torch.randn(B * H * W, K, device='cuda').bfloat16()
        w2t = (torch.randn(C, K, device='cuda') / K ** 0.5).bfloat16()
        keep = torch.full((B,), 1.25, device='cuda')
        w9c = (torch.randn(9, C, device='cuda') / 3).contiguous()
        b = 0.1 * torch.randn(C, device='cuda')

        def two():
            da2 = ops.gemm_nt(dout, w2t, rowscale=keep, rows_per_scale=H * W)
            return ops.dwconv3_bwd_fused(da2, x, w9c, b, B, H, W, act=1)
        t2 = timed(two)
        t1 = timed(lambda: ops.dwconv3_bwd_fc2(dout, w2t, keep, x, w9c, b, B, H, W))
        print(f'{B}x{H}x{W}x{C} K={K}: GEMM + depthwise backward {t2:8.1f} us   one kernel {t1:8.1f} us', flush=True)


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'fc2':
    fc2()
elif __name__ == '__main__':
    main()
