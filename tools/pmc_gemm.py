"""A handful of GEMM launches (one warm-up + two measured per shape) for rocprofv3 --pmc passes.
    python3 tools/pmc_gemm.py                      the default list
    python3 tools/pmc_gemm.py nt M N K [res]       one NT shape        python3 tools/pmc_gemm.py tn R M N     one TN shape"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd import ops

NT = [(262144, 512, 512), (262144, 512, 128), (65536, 1280, 320), (1048576, 256, 64), (262144, 2560, 512), (262144, 512, 2560)]
TN = [(262144, 512, 512), (65536, 320, 1280)]
res = False
if len(sys.argv) > 1:
    kind, dims = sys.argv[1], tuple(int(v) for v in sys.argv[2:5])
    res = 'res' in sys.argv
    NT, TN = ([dims], []) if kind == 'nt' else ([], [dims])
for M, N, K in NT:
    a = torch.randn(M, K, device='cuda').bfloat16(); b = torch.randn(N, K, device='cuda').bfloat16()
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    r = torch.randn(M, N, device='cuda').bfloat16() if res else None
    for _ in range(3):
        ops.gemm_nt(a, b, out=out, res=r)
    torch.cuda.synchronize()
    del a, b, out
for R, M, N in TN:
    a = torch.randn(R, M, device='cuda').bfloat16(); b = torch.randn(R, N, device='cuda').bfloat16()
    for _ in range(3):
        ops.gemm_tn(a, b)
    torch.cuda.synchronize()
    del a, b
print('done')
