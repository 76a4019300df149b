import os, sys, torch
sys.path.insert(0, '/root/repo')
from semseg_amd import ops
M, N, K = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (262144, 512, 2560))]
a = torch.randn(M, K, device='cuda').bfloat16(); b = torch.randn(N, K, device='cuda').bfloat16(); out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
for _ in range(3): ops.gemm_nt(a, b, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.gemm_nt(a, b, out=out)
e1.record(); torch.cuda.synchronize()
print('SS_GEMM_DBG', os.environ.get('SS_GEMM_DBG'), f'{e0.elapsed_time(e1) * 100:.1f} us')
