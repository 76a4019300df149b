"""Times the BatchNorm-of-a-resize entry points (ss_bn_*_up) at the headline shape of LawinHead's `cat` (16 x 128^2 -> 256^2, 512
channels, bf16) next to the unfused kernels they replace.  Run under `rocprofv3 --kernel-trace --stats` for per-kernel times."""
import sys
import torch
sys.path.insert(0, '.')
from semseg_amd import ops

B, hi, wi, C = 16, 128, 128, 512
ho, wo = 2 * hi, 2 * wi
T = torch.bfloat16
torch.manual_seed(0)
x = torch.randn(B * hi * wi, C, device='cuda').to(T)
wide = torch.randn(B * ho * wo, C + 48, device='cuda').to(T)
dy = wide[:, :C]
g, b = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
out = torch.empty(B * ho * wo, C + 48, dtype=T, device='cuda')


def timed(name, fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f'{name:28s} {e0.elapsed_time(e1) / n * 1e3:9.1f} us')


st = ops.bn_stats_up(x, B, hi, wi, ho, wo)
z = ops.bilinear_fwd(x, B, hi, wi, ho, wo)
timed('bilinear_fwd', lambda: ops.bilinear_fwd(x, B, hi, wi, ho, wo))
timed('bn_stats', lambda: ops.bn_stats(z))
timed('bn_apply', lambda: ops.bn_apply(z, st, g, b, 1, out=out[:, :C]))
timed('bn_bwd', lambda: ops.bn_bwd(dy, z, st, g, b, 1))
dz, _ = ops.bn_bwd(dy, z, st, g, b, 1)
timed('bilinear_bwd', lambda: ops.bilinear_bwd(dz, B, hi, wi, ho, wo))
timed('bn_stats_up', lambda: ops.bn_stats_up(x, B, hi, wi, ho, wo))
timed('bn_apply_up', lambda: ops.bn_apply_up(x, B, hi, wi, ho, wo, st, g, b, 1, out=out[:, :C]))
timed('bn_bwd_up', lambda: ops.bn_bwd_up(dy, x, B, hi, wi, ho, wo, st, g, b, 1))
timed('bn_bwd_up_to_input', lambda: ops.bn_bwd_up_to_input(dy, x, B, hi, wi, ho, wo, st, g, b, 1))
