"""Timings of the projection GEMM (ss_gemm_nt) on the shapes of the headline step.  SS_GEMM_WIDE=0 / 1 / 2 and SS_GEMM_BIG=0 / 1
select the 256 x 128 and 256 x 256 tile kernels for A/B comparisons."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd import ops
NT = [(262144, 512, 512, 0), (1048576, 256, 64, 0), (262144, 512, 128, 0), (262144, 512, 256, 1), (262144, 512, 256, 0), (65536, 1280, 320, 0),
      (262144, 2560, 512, 0), (262144, 512, 2560, 0), (262144, 256, 512, 0), (262144, 128, 128, 1), (262144, 128, 512, 1), (262144, 128, 512, 0),
      (262144, 1536, 512, 0), (262144, 512, 1536, 0), (16384, 2048, 512, 0), (16384, 512, 2048, 1), (16384, 512, 512, 0), (65536, 320, 1280, 1),
      (1048576, 128, 64, 0), (262144, 128, 128, 0), (16384, 1280, 320, 0), (1048576, 512, 576, 0)]
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tag = f'wide={os.environ.get("SS_GEMM_WIDE", "1")} big={os.environ.get("SS_GEMM_BIG", "1")}'
for M, N, K, r in NT:
    a = torch.randn(M, K, device='cuda').bfloat16(); b = torch.randn(N, K, device='cuda').bfloat16()
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16); res = torch.randn(M, N, device='cuda').bfloat16() if r else None
    ms = timeit(lambda: ops.gemm_nt(a, b, out=out, res=res))
    byt = 2 * (M * K + N * K + M * N * (2 if r else 1))
    print(f'{tag} nt M={M:8d} N={N:5d} K={K:5d} res={r} {ms*1e3:8.1f} us {2*M*N*K/ms/1e9:7.1f} TFLOP/s {byt/ms/1e6:6.0f} GB/s', flush=True)
    del a, b, out, res
