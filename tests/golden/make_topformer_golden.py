"""Golden vectors / manifests for TopFormerModel from the REFERENCE (imported from /root/reference, CPU fp32, eval mode).
The reference's TopFormer can be constructed once per process, so this script builds ONE model per run:

    cd /root/repo && for a in B:simple B:refiner S:simple T:simple T:simple:concat; do PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_topformer_golden.py $a; done

(alias:head[:fusion]; fusion defaults to 'sum' and is part of the file name only when given.)
"""
import json
import os
import sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(1, '/root/reference')
sys.dont_write_bytecode = True
from semseg_amd.detinit import det_fill_, det_input  # noqa: E402
from core.models import TopFormerModel  # noqa: E402  (reference)
from modules.topformer import TopFormerBackBone  # noqa: E402  (reference)

alias, head, *rest = sys.argv[1].split(':')
fusion = rest[0] if rest else 'sum'
tag = f'{alias}_{head}' + (f'_{fusion}' if rest else '')
# core/models.py:46-49 sizes the head for the model-level fusion but does not forward it: the head needs the same mode in its own cfg
head_cfg = {'alias': head, **({'fusion': fusion} if rest else {})}
cfgs = getattr(TopFormerBackBone, '_TopFormerBackBone__available_configs')
tables = {}
for a, c in cfgs.items():
    tables[a] = {'stages': [[[b['kernel_size'], b['expand_ratio'], b['base_channels'], b['stride']] for b in st.values()]
                            for st in c['stage_configs'].values()],
                 'out_channels': list(c['out_channels'].values()), 'num_heads': c['num_heads'], 'c2t_stride': c['c2t_stride'],
                 'depths': c['depths'], 'key_dim': c['key_dim'], 'attn_ratios': c['attn_ratios'], 'mlp_ratios': c['mlp_ratios']}
json.dump(tables, open(os.path.join(HERE, 'topformer_configs.json'), 'w'), indent=1)

net = det_fill_(TopFormerModel(6, alias, 3, 'dot_sum', fusion, frozenset(head_cfg.items()))).eval()
manifest = {k: list(v.shape) for k, v in net.state_dict().items()}
json.dump(manifest, open(os.path.join(HERE, f'topformer_{tag}_manifest.json'), 'w'), indent=0)
x = det_input('top.x', (2, 3, 128, 192))
with torch.no_grad():
    feats = net.backbone(x)
    out = net(x)
save = {'logits': out.numpy()}
for k, v in feats.items():                      # the two coarsest injected maps (the fine ones would be megabytes)
    if k in ('stage_3', 'stage_4'):
        save['feat.' + k] = v.numpy()
np.savez_compressed(os.path.join(HERE, f'topformer_{tag}.npz'), **save)
print(tag, 'logits', tuple(out.shape), 'params', sum(p.numel() for p in net.parameters()), 'keys', len(manifest))

# ---- one training step (append `train` as the second argument): train-mode forward (batch-statistic BatchNorm), OHEM cross-entropy as
# train.py:61-67 configures it, backward; logits, loss, a sample of gradients from every part of the network and the BatchNorm running
# statistics after the step.  Written to topformer_<tag>_train.npz.
#
# At this size the transformer part holds 2 x 2 x 3 = 12 token rows, so its train-mode BatchNorms divide by the spread of 12 samples and
# the step is poorly conditioned: the reference's OWN fp32 forward is 0.7 - 1.2e-4 away from the same reference run in fp64
# (topformer_train_sensitivity.py).  The golden values are therefore the reference in FLOAT64 (stored as fp32), and beside each of them
# the file records how far the reference's fp32 run is from it (`ref32err.*`): the test holds the MI355X path to that yardstick.
if len(sys.argv) > 2 and sys.argv[2] == 'train':
    from semseg_amd.detinit import det_labels  # noqa: E402
    from loss.seg_loss import OhemCrossEntropyLoss  # noqa: E402  (reference)
    net.train()
    xt = det_input('top.xt', (2, 3, 128, 192))
    lab = det_labels('top.lab', (2, 128, 192), 6, cell=8)
    pick_parts = ('stem.', 'stage_1.resblock_2.', 'stage_4.resblock_1.', 'extractor_1.', 'extractor_4.mlp', 'post_stages.stage_1.', 'post_stages.stage_4.',
                  'head.', 'classifier.')

    def one_step(model, x):
        for p in model.parameters():
            p.grad = None
        logits = model(x)
        loss = OhemCrossEntropyLoss(ignore_index=0, threshold=0.7, min_kept=0.1)(logits, lab)
        loss.backward()
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        return logits.detach().clone(), loss.detach().clone(), grads

    lg32, ls32, g32 = one_step(net, xt)
    missing = [n for n, p in net.named_parameters() if p.grad is None]
    tsave = {}
    for k, v in net.state_dict().items():                                          # buffers after ONE (the fp32) step
        if k.endswith('running_mean') or k.endswith('running_var') or k.endswith('num_batches_tracked'):
            if any(s in k for s in ('stem.', 'stage_2.resblock_1.', 'extractor_2.', 'post_stages.stage_3.', 'head.')):
                tsave['after.' + k] = v.numpy().copy()
    net.double()
    lg64, ls64, g64 = one_step(net, xt.double())
    relmax = lambda a, b: float((a.double() - b).abs().max() / max(float(b.abs().max()), 1e-300))
    names = list(g64.keys())
    tsave['logits'], tsave['loss'] = lg64.float().numpy(), ls64.numpy()
    tsave['ref32err.logits'], tsave['ref32err.loss'] = np.float64(relmax(lg32, lg64)), np.float64(abs(float(ls32) - float(ls64)) / abs(float(ls64)))
    # every parameter's gradient as its L2 norm (cheap, catches a wrong chain anywhere) and a sample of full tensors
    tsave['gradnorm.names'] = np.array(names)
    tsave['gradnorm.values'] = np.array([float(g64[n].norm()) for n in names], dtype=np.float64)
    tsave['ref32err.gradnorm'] = np.array([abs(float(g32[n].double().norm()) - float(g64[n].norm())) for n in names], dtype=np.float64)   # absolute
    pick = [n for n in names if any(s in n for s in pick_parts) and g64[n].numel() <= 40000]
    for n in pick:
        tsave['grad.' + n] = g64[n].float().numpy()
        tsave['ref32err.grad.' + n] = np.float64(relmax(g32[n], g64[n]))
    cat = lambda d: torch.cat([d[n].double().flatten() for n in pick])
    tsave['ref32err.grad_l2_all'] = np.float64(float((cat(g32) - cat(g64)).norm() / cat(g64).norm()))
    np.savez_compressed(os.path.join(HERE, f'topformer_{tag}_train.npz'), **tsave)
    errs = [float(tsave['ref32err.grad.' + n]) for n in pick]
    print(tag, 'train: loss', float(ls64), 'params with grad', len(names), 'without', missing, 'full gradients kept', len(pick),
          'buffers', sum(k.startswith('after.') for k in tsave))
    print(tag, 'reference fp32 vs fp64: logits', float(tsave['ref32err.logits']), 'loss', float(tsave['ref32err.loss']), 'sampled gradients: median',
          float(np.median(errs)), 'max', float(np.max(errs)), 'all sampled gradients as one vector (rel-L2)', float(tsave['ref32err.grad_l2_all']))
