"""The 256 x 256 NT kernel's forms side by side, one process per form (SS_GEMM_BIG is read once): 1 = one tile per workgroup, two stages;
2 = persistent; 3 = one tile per workgroup, three A stages + two B stages."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, ROOT)
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
    for M, N, K in [(262144, 512, 512), (65536, 1280, 320), (262144, 2048, 512), (262144, 512, 2048), (262144, 512, 256), (262144, 256, 512),
                    (1048576, 512, 576), (1048576, 256, 512), (16384, 2048, 512), (16384, 512, 512), (16384, 512, 2048), (262144, 512, 128)]:
        a = torch.randn(M, K, device='cuda').bfloat16(); b = torch.randn(N, K, device='cuda').bfloat16()
        out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        ms = timeit(lambda: ops.gemm_nt(a, b, out=out))
        ref = (a[:64].float() @ b.float().t())
        err = float((out[:64].float() - ref).abs().max() / ref.abs().max())
        print(f'{M} {N} {K} {ms * 1e3:.1f} {err:.1e}', flush=True)
    sys.exit(0)
res = {}
for mode in ('1', '2', '3'):
    env = dict(os.environ, SS_GEMM_BIG=mode)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=env, capture_output=True, text=True).stdout
    for l in out.splitlines():
        f = l.split()
        if len(f) == 5:
            res.setdefault(tuple(f[:3]), {})[mode] = (float(f[3]), f[4])
for k, v in res.items():
    M, N, K = map(int, k)
    print(f'nt M={M:8d} N={N:5d} K={K:5d}  ' + '  '.join(f'[{m}] {t:7.1f} us {2 * M * N * K / t / 1e6:5.0f} TF err {e}' for m, (t, e) in sorted(v.items())), flush=True)
