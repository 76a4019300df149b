import os, sys, torch
sys.path.insert(0, '/root/repo')
from semseg_amd import ops
torch.manual_seed(0)
for (M, N, K) in [(12288, 16, 16), (12288, 8, 8), (12288, 32, 8), (12288, 8, 32), (12288, 16, 8)]:
    a = torch.randn(M, K, device='cuda'); b = torch.randn(N, K, device='cuda'); bias = torch.randn(N, device='cuda')
    ref = (a.double() @ b.double().t() + bias.double())
    for v in ('0', '1'):
        os.environ['SS_GEMM_ROWS'] = v
        out = ops.gemm_nt(a, b, bias=bias)
        e = ((out.double() - ref).abs().max() / ref.abs().max()).item()
        print(f'nt fp32 M={M} N={N} K={K} rows={v}: max rel err {e:.2e}')
for (R, N) in [(12288, 8), (12288, 16), (12288, 32), (300000, 8)]:
    a = torch.randn(R, 8, device='cuda'); b = torch.randn(R, N, device='cuda')
    ref = a.double().t() @ b.double()
    for v in ('0', '1'):
        os.environ['SS_GEMM_ROWS'] = v
        out = ops.gemm_tn(a, b)
        e = ((out.double() - ref).abs().max() / ref.abs().max()).item()
        print(f'tn fp32 R={R} N={N} rows={v}: max rel err {e:.2e}')
