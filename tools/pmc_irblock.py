"""The fused inverted-residual kernel at the stage-1 / stage-2 shapes of TopFormer-B (B = 4, 2048 x 2048), for timing and rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd import ops
from semseg_amd.topformer import InvertedResidual, _Plan, auto_pad
from semseg_amd.detinit import det_fill_

T = torch.bfloat16
SHAPES = [(16, 1, 16, 3, 1, 1024), (16, 4, 32, 3, 2, 1024), (32, 3, 32, 3, 1, 512), (32, 3, 64, 5, 2, 512)]
only = int(sys.argv[1]) if len(sys.argv) > 1 else -1
for i, (cin, e, cout, k, s, hw) in enumerate(SHAPES):
    if only >= 0 and i != only:
        continue
    blk = det_fill_(InvertedResidual(cin, cout, k, s, e)).cuda().eval()
    f = _Plan._fused_block(blk, T)
    B, H, W = 4, hw, hw
    x = torch.randn(B * H * W, cin, device='cuda').to(T)
    p0, _ = auto_pad(k, s)
    run = lambda: ops.ir_block_fwd(x, B, H, W, cout, k, s, p0, p0, H // s, W // s, f['w_in'], f['b_in'], f['act_in'], f['w_dw'], f['b_dw'],
                                   f['act_dw'], f['w_out'], f['b_out'], residual=blk.residual_connect)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record(); torch.cuda.synchronize()
    io = B * H * W * cin * 2 + B * (H // s) * (W // s) * cout * 2
    t = e0.elapsed_time(e1) / 5
    print(f'block {i}: {cin}->{cin * e}->{cout} k{k} s{s} at {hw}^2: {t * 1e3:7.1f} us  {io / t / 1e6:6.0f} GB/s of in + out', flush=True)
