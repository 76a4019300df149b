"""Ablation timings of the 256-tile NT GEMM (SS_GEMM_DBG bit 1: no C stores, 2: no operand loads after tile 0, 4: no MFMA)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd import ops
NT = [(262144, 512, 512), (65536, 1280, 320), (262144, 2560, 512), (262144, 512, 2560)]
if os.environ.get('SS_SHAPES'):
    NT = [tuple(int(v) for v in t.split('x')) for t in os.environ['SS_SHAPES'].split(',')]
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
    print(f'dbg={os.environ.get("SS_GEMM_DBG","0")} nt M={M:8d} N={N:5d} K={K:5d} {ms*1e3:8.1f} us {2*M*N*K/ms/1e9:7.1f} TFLOP/s', flush=True)
