import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
os.chdir('/root/repo')
import test_topformer_train_gpu as T
gd = os.path.join('/root/repo/tests/golden')
for tag in sorted(T.VARIANTS):
    g = np.load(os.path.join(gd, f'topformer_{tag}_train.npz'))
    net = T.make(tag).cuda().train()
    logits, loss = T._step(net)
    params = dict(net.named_parameters())
    names, norms, nerr = [str(n) for n in g['gradnorm.names']], g['gradnorm.values'], g['ref32err.gradnorm']
    agg = float(g['ref32err.grad_l2_all'])
    rows = []
    for n, want, ref_dev in zip(names, norms, nerr):
        got = float(params[n].grad.double().norm())
        rows.append((abs(got - want) / max(want, 1e-30), ref_dev / max(want, 1e-30), n, want))
    rows = [r for r in rows if r[3] > 1e-6 * float(np.max(norms))]
    rel = np.array([r[0] for r in rows]); refrel = np.array([r[1] for r in rows])
    print(tag, 'params', len(rows), 'agg ref32err.grad_l2_all', agg, 'ours: median rel dev', np.median(rel), 'p90', np.percentile(rel, 90), 'max', rel.max(),
          '| reference fp32: median', np.median(refrel), 'p90', np.percentile(refrel, 90), 'max', refrel.max())
    for r in sorted(rows, reverse=True)[:5]:
        print('   ours %.4f  ref32 %.4f  %s' % (r[0], r[1], r[2]))
