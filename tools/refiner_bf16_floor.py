import sys, torch
sys.path.insert(0, '/root/repo')
from semseg_amd.topformer import TopFormerModel
from semseg_amd.detinit import det_fill_
from semseg_amd.runtime import set_compute_dtype
from semseg_amd import ops
net = det_fill_(TopFormerModel(6, 'B', head_cfg=frozenset({'alias': 'refiner'}.items()))).cuda().eval()
g = torch.Generator().manual_seed(7)
for S in (256, 512, 1024, 2048):
    x = (0.8 * torch.rand(1, 3, S // 64, S // 64, generator=g).repeat_interleave(64, 2).repeat_interleave(64, 3) + 0.2 * torch.rand(1, 3, S, S, generator=g)).cuda()
    def l2(a, b): return float((a.float() - b.float()).norm() / b.float().norm())
    with torch.no_grad():
        ref = net(x)
        plan = net.backbone._plan_for(net, x)
        feats = list(net.backbone._run(x, plan).values())
        h4, w4 = feats[0][1], feats[0][2]
        rounded = [(t.bfloat16().float(), h, w) for t, h, w in feats]
        lg = net._refiner(plan, rounded, 1, h4, w4)
        full = ops.bilinear_fwd(lg, 1, h4, w4, S, S).view(1, S, S, -1)[..., :6].permute(0, 3, 1, 2)
        set_compute_dtype(torch.bfloat16)
        out = net(x)
        planb = net.backbone._plan_for(net, x)
        featsb = list(net.backbone._run(x, planb).values())
        set_compute_dtype(None)
        ferr = [l2(fb[0], f[0]) for fb, f in zip(featsb, feats)]
    print(S, 'one rounding of the head inputs:', l2(full, ref), ' bf16 engine:', l2(out, ref), ' stage feature errors:', ferr, flush=True)
