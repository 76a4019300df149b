import os, sys
sys.path.insert(0, os.getcwd())
import torch
from semseg_amd import ops
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for M, N, K in [(262144, 512, 128), (1048576, 128, 128), (524288, 256, 128), (262144, 512, 64), (262144, 512, 256), (2097152, 64, 128), (262144, 1024, 128)]:
    a = torch.randn(M, K, device='cuda').bfloat16(); b = torch.randn(N, K, device='cuda').bfloat16()
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    ms = timeit(lambda: ops.gemm_nt(a, b, out=out))
    byt = 2 * (M * K + N * K + M * N)
    print(f'nt M={M:8d} N={N:5d} K={K:5d} {ms*1e3:8.1f} us {2*M*N*K/ms/1e9:7.1f} TFLOP/s {byt/ms/1e6:6.0f} GB/s', flush=True)
    # the same output bytes written by a plain copy kernel for reference
    src = torch.empty_like(out)
    ms2 = timeit(lambda: out.copy_(src))
    print(f'     copy of the output alone: {ms2*1e3:8.1f} us')
    del a, b, out, src
