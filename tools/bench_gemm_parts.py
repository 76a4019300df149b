"""Decomposition of the one-tile-per-workgroup 256 x 256 NT kernel by its debug switches (SS_GEMM_BIG=1; SS_GEMM_DBG bit 1: no stores,
2: no operand fills after the first, 4: no MFMA)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['SS_GEMM_BIG'] = '1'           # the two-stage form carries the switches
import torch
from semseg_amd import ops


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for M, N, K in [(262144, 512, 512), (65536, 1280, 320), (262144, 2048, 512), (262144, 512, 2048)]:
    a = torch.randn(M, K, device='cuda').bfloat16(); b = torch.randn(N, K, device='cuda').bfloat16()
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    line = f'nt M={M:8d} N={N:5d} K={K:5d}'
    for dbg in ('0', '1', '2', '4', '3', '5', '6', '7'):
        os.environ['SS_GEMM_DBG'] = dbg
        ms = timeit(lambda: ops.gemm_nt(a, b, out=out))
        line += f'  [{dbg}] {ms * 1e3:6.1f}'
    os.environ['SS_GEMM_DBG'] = '0'
    print(line, flush=True)
