"""Timings of the weight-gradient GEMM (ss_gemm_tn / ss_gemm_tn_colsum) on the shapes of the headline step.  SS_GEMM_TN_BIG=0
disables the 256-tile kernel for an A/B comparison."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd import ops
TN = [(512, 2560, 262144, 0), (512, 560, 1048576, 0), (512, 512, 262144, 1), (512, 1536, 262144, 0), (256, 512, 262144, 1), (512, 256, 262144, 0),
      (1280, 320, 65536, 1), (320, 1280, 65536, 1), (320, 320, 65536, 1), (2048, 512, 16384, 1), (512, 2048, 16384, 1), (512, 512, 16384, 1),
      (1024, 512, 16384, 1), (512, 2880, 16384, 1), (64, 256, 1048576, 1), (256, 64, 1048576, 1), (128, 128, 262144, 1), (512, 128, 262144, 1),
      (128, 512, 262144, 1), (64, 64, 1048576, 1), (640, 320, 16384, 1)]
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for M, N, R, cs in TN:
    a = torch.randn(R, M, device='cuda').bfloat16(); b = torch.randn(R, N, device='cuda').bfloat16()
    out = torch.empty(M, N, device='cuda')
    ms = timeit(lambda: ops.gemm_tn(a, b, out=out, colsum=bool(cs)))
    byt = 2 * R * (M + N)
    print(f'big={os.environ.get("SS_GEMM_TN_BIG","1")} tn M={M:5d} N={N:5d} R={R:8d} cs={cs} {ms*1e3:8.1f} us {2*M*N*R/ms/1e9:7.1f} TFLOP/s {byt/ms/1e6:7.0f} GB/s', flush=True)
    del a, b
