"""GEMM micro-benchmark over the shapes the MiT-B2 + Lawin(512) step produces (B=16, 1024x1024)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd import ops

NT = [(262144, 512, 128), (65536, 1280, 320), (262144, 512, 512), (1048576, 256, 64), (1048576, 560, 512), (262144, 2560, 512),
      (1048576, 512, 560), (1048576, 64, 64), (65536, 320, 320), (262144, 256, 512), (262144, 128, 128), (65536, 320, 1280),
      (262144, 512, 2560), (1048576, 64, 256), (16384, 2048, 512), (16384, 512, 2048)]
TN = [(262144, 128, 128), (1048576, 64, 64), (1048576, 512, 560), (262144, 512, 2560), (1048576, 64, 256), (65536, 320, 320),
      (65536, 320, 1280), (262144, 512, 512), (16384, 512, 2048)]


YARD = '--yardstick' in sys.argv

if YARD:            # what a plain device copy reaches on this box (read + write bytes per second)
    for mb in (64, 256, 1024):
        src = torch.empty(mb << 20, dtype=torch.uint8, device='cuda'); dst = torch.empty_like(src)
        dst.copy_(src); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dst.copy_(src)
        e1.record(); torch.cuda.synchronize()
        print(f'copy {mb} MiB: {2 * (mb << 20) * 10 / e0.elapsed_time(e1) / 1e6:7.0f} GB/s (read + write)', flush=True)
        del src, dst


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for M, N, K in NT:
    a = torch.randn(M, K, device='cuda').bfloat16(); b = torch.randn(N, K, device='cuda').bfloat16()
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    ms = timeit(lambda: ops.gemm_nt(a, b, out=out))
    byt = (M * K + N * K + M * N) * 2
    lib = timeit(lambda: torch.mm(a, b.t(), out=out)) if YARD else 0.0      # yardstick only: the vendor library on the same shape
    ideal = max(2 * M * N * K / 2.5e15, byt / 8e12) * 1e3
    print(f'nt M={M:8d} N={N:5d} K={K:5d} {ms*1e3:8.1f} us {2*M*N*K/ms/1e9:7.1f} TFLOP/s {byt/ms/1e6:7.0f} GB/s  frac {ideal/ms:.2f}'
          + (f'   [library {lib*1e3:8.1f} us]' if YARD else ''), flush=True)
    del a, b, out
for R, M, N in TN:
    a = torch.randn(R, M, device='cuda').bfloat16(); b = torch.randn(R, N, device='cuda').bfloat16()
    ms = timeit(lambda: ops.gemm_tn(a, b))
    byt = (R * M + R * N) * 2 + M * N * 4
    lib = timeit(lambda: torch.mm(a.t(), b)) if YARD else 0.0
    ideal = max(2 * M * N * R / 2.5e15, byt / 8e12) * 1e3
    print(f'tn R={R:8d} M={M:5d} N={N:5d} {ms*1e3:8.1f} us {2*M*N*R/ms/1e9:7.1f} TFLOP/s {byt/ms/1e6:7.0f} GB/s  frac {ideal/ms:.2f}'
          + (f'   [library {lib*1e3:8.1f} us]' if YARD else ''), flush=True)
    del a, b
