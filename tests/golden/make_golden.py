"""Generate the golden vectors under tests/golden/ by running the REFERENCE
(imported from /root/reference, CPU, fp32) and, in the same pass, check the
oracle restatement (oracle/) against it.  Runs only in the build container
(the reference never travels to the GPU box); the .npz outputs are committed.

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import os
import sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(1, '/root/reference')
sys.dont_write_bytecode = True
torch.manual_seed(0)
torch.set_num_threads(8)

from semseg_amd.detinit import det_fill_, det_input, det_labels  # noqa: E402
from oracle import mit as omit, lawin as olawin, losses as olosses, convnext as oconvnext  # noqa: E402
from modules.segformer import SegFormer  # noqa: E402  (reference)
from modules.lawin import LawinHead  # noqa: E402  (reference)
from loss.seg_loss import OhemCrossEntropyLoss, DiceLoss, dice_loss  # noqa: E402  (reference)
from helper.weights import class_weights  # noqa: E402  (reference)
import torch.nn.functional as F  # noqa: E402


def rel(a, b):
    return float((a.detach() - b.detach()).abs().max() / (b.detach().abs().max() + 1e-12))


def np_(t):
    return t.detach().cpu().numpy()


DEVS = {}                # oracle-vs-reference deviation per checked tensor: two fp32 CPU evaluation orders of the same maths, i.e.
                         # the fp32 noise floor of that tensor; stored in the fixtures as 'dev.<name>' (GPU tests may widen the
                         # 1e-4 gradient bar to 2x this value where a chain of training-mode BatchNorms amplifies rounding)


def f64dev(save, name, v32, v64):
    """'f64dev.<name>' = deviation of the reference's OWN fp32 result from its float64 evaluation of the same graph: how far fp32
    rounding alone moves this tensor.  A GPU fp32 result cannot be asked to sit closer to the fp32 reference than that."""
    save['f64dev.' + name] = np.float32(rel(v32.double(), v64))


def devs(prefix):
    return {'dev.' + k[len(prefix):]: np.float32(v) for k, v in DEVS.items() if k.startswith(prefix)}


def check(name, a, b, tol=2e-5):
    r = rel(a, b)
    DEVS[name] = r
    print(f'  oracle-vs-reference {name}: rel {r:.2e}')
    # analytically-zero quantities (e.g. a bias in front of a training-mode BN) are pure rounding noise
    assert r < tol or float((a - b).abs().max()) < 1e-6, (name, r)


GRAD_KEYS_ENC = ['patch_embed1.proj.weight', 'patch_embed1.norm.weight', 'block1.0.attn.q.weight',
                 'block1.0.attn.sr.weight', 'block1.0.attn.norm.bias', 'block1.1.attn.kv.weight',
                 'block1.0.mlp.dw_conv.dw_conv.weight', 'block1.0.mlp.dw_conv.dw_conv.bias',
                 'block2.1.mlp.fc2.weight', 'block3.0.attn.proj.bias', 'block4.1.attn.kv.bias',
                 'block4.0.mlp.fc1.weight', 'patch_embed3.proj.weight', 'norm2.weight', 'norm4.bias']
GRAD_KEYS_HEAD = ['linear_c1.proj.weight', 'linear_c3.proj.weight', 'linear_fuse.conv.weight',
                  'linear_fuse.bn.weight', 'lawin_8.position_mixing.3.weight', 'lawin_8.position_mixing.63.bias',
                  'lawin_4.position_mixing.0.weight', 'lawin_2.position_mixing.3.bias',
                  'lawin_8.g.weight', 'lawin_4.theta.bias', 'lawin_2.phi.weight', 'lawin_8.conv_out.0.weight',
                  'lawin_4.conv_out.1.bias', 'image_pool.1.conv.weight', 'image_pool.1.bn.weight',
                  'cat.conv.weight', 'cat.bn.bias', 'low_level_fuse.conv.weight', 'linear_pred.weight',
                  'linear_pred.bias']


def detached_sd(module):
    return {k: v.detach().clone() for k, v in module.state_dict().items()}


def grad_sd(sd):
    out = {}
    for k, v in sd.items():
        out[k] = v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v.clone()
    return out


def gen_mit():
    enc = det_fill_(SegFormer('B0')).eval()
    x = det_input('mit.x', (2, 3, 64, 64))
    outs = enc(x)
    projs = [det_input(f'mit.proj{i}', o.shape, -1, 1) for i, o in enumerate(outs)]
    loss = sum((o * p).sum() for o, p in zip(outs, projs))
    loss.backward()
    grads = {k: p.grad for k, p in enc.named_parameters()}
    # oracle
    sd = grad_sd(detached_sd(enc))
    oouts = omit.mit_forward(sd, x, 'B0')
    sum((o * p).sum() for o, p in zip(oouts, projs)).backward()
    for i in range(4):
        check(f'mit.out{i}', oouts[i], outs[i])
    for k in GRAD_KEYS_ENC:
        check(f'mit.grad.{k}', sd[k].grad, grads[k], 5e-5)
    save = {'x': np_(x)}
    for i in range(4):
        save[f'out{i}'] = np_(outs[i])
    for k in GRAD_KEYS_ENC:
        save['grad.' + k] = np_(grads[k])
    enc64 = det_fill_(SegFormer('B0')).double().eval()
    outs64 = enc64(x.double())
    sum((o * p.double()).sum() for o, p in zip(outs64, projs)).backward()
    g64 = {k: p.grad for k, p in enc64.named_parameters()}
    for i in range(4):
        f64dev(save, f'out{i}', outs[i], outs64[i])
    for k in GRAD_KEYS_ENC:
        f64dev(save, 'grad.' + k, grads[k], g64[k])
    # the reference's OWN bf16 path (torch.autocast), as gen_head: the noise floor a bf16 engine is measured against
    enc_b = det_fill_(SegFormer('B0')).eval()
    with torch.autocast('cpu', dtype=torch.bfloat16):
        outs_b = enc_b(x)
    sum((o.float() * p).sum() for o, p in zip(outs_b, projs)).backward()
    gb = {k: p.grad for k, p in enc_b.named_parameters()}
    for k in GRAD_KEYS_ENC:
        save['bf16.grad.' + k] = np_(gb[k])
    save.update(devs('mit.'))
    np.savez_compressed(os.path.join(HERE, 'mit_b0_64.npz'), **save)


def gen_head(training):
    tag = 'train' if training else 'eval'
    chans = [32, 64, 160, 256]
    head = det_fill_(LawinHead(chans, 128, 6))
    head.train(training)
    head.dropout.p = 0.0          # RNG-free fixture; Dropout2d is covered through explicit masks
    feats = [det_input(f'head.f{i}', (2, c, 32 >> i, 32 >> i), -1, 1).requires_grad_(True)
             for i, c in enumerate(chans)]
    sd0 = detached_sd(head)
    out = head(feats)
    proj = det_input('head.proj', out.shape, -1, 1)
    (out * proj).sum().backward()
    grads = {k: p.grad for k, p in head.named_parameters()}
    # oracle
    sd = grad_sd(sd0)
    ofe = [f.detach().clone().requires_grad_(True) for f in feats]
    oout = olawin.lawin_head(sd, ofe, training=training)
    (oout * proj).sum().backward()
    check(f'head.{tag}.out', oout, out)
    for i in range(4):
        check(f'head.{tag}.dfeat{i}', ofe[i].grad, feats[i].grad, 5e-5)
    for k in GRAD_KEYS_HEAD:
        check(f'head.{tag}.grad.{k}', sd[k].grad, grads[k], 5e-5)
    save = {'out': np_(out)}
    for i in range(4):
        save[f'dfeat{i}'] = np_(feats[i].grad)
    for k in GRAD_KEYS_HEAD:
        save['grad.' + k] = np_(grads[k])
    head64 = det_fill_(LawinHead(chans, 128, 6)).double()
    head64.train(training)
    head64.dropout.p = 0.0
    f64 = [f.detach().double().requires_grad_(True) for f in feats]
    out64 = head64(f64)
    (out64 * proj.double()).sum().backward()
    g64 = {k: p.grad for k, p in head64.named_parameters()}
    f64dev(save, 'out', out, out64)
    for i in range(4):
        f64dev(save, f'dfeat{i}', feats[i].grad, f64[i].grad)
    for k in GRAD_KEYS_HEAD:
        f64dev(save, 'grad.' + k, grads[k], g64[k])
    if training:
        # the reference's OWN bf16 path (torch.autocast) on the same inputs: its deviation from fp32 is the noise floor the
        # bf16 GPU tests measure our deviation against (a 2e-2 budget that the reference itself misses cannot be a bar)
        head_b = det_fill_(LawinHead(chans, 128, 6)).train()
        head_b.dropout.p = 0.0
        fb = [f.detach().clone().requires_grad_(True) for f in feats]
        with torch.autocast('cpu', dtype=torch.bfloat16):
            ob = head_b(fb)
        (ob.float() * proj).sum().backward()
        gb = {k: p.grad for k, p in head_b.named_parameters()}
        save['bf16.out'] = np_(ob.float())
        for i in range(4):
            save[f'bf16.dfeat{i}'] = np_(fb[i].grad)
        for k in GRAD_KEYS_HEAD:
            save['bf16.grad.' + k] = np_(gb[k])
        after = head.state_dict()
        for k in ('linear_fuse.bn.running_mean', 'linear_fuse.bn.running_var', 'lawin_8.conv_out.1.running_var',
                  'image_pool.1.bn.running_mean', 'image_pool.1.bn.running_var', 'cat.bn.running_var',
                  'cat.bn.num_batches_tracked'):
            check(f'head.train.{k}', sd[k].float(), after[k].float())
            save['after.' + k] = np_(after[k])
    save.update(devs(f'head.{tag}.'))
    np.savez_compressed(os.path.join(HERE, f'lawin_e128_{tag}.npz'), **save)


def gen_head_e256():
    """LawinHead with embed_dim 256 (SURVEY 8c asks for E = 128 and 256), eval mode, 2 x 128 x 128 tiles: logits and feature gradients."""
    chans = [32, 64, 160, 256]
    head = det_fill_(LawinHead(chans, 256, 6)).eval()
    feats = [det_input(f'head256.f{i}', (2, c, 32 >> i, 32 >> i), -1, 1).requires_grad_(True) for i, c in enumerate(chans)]
    sd = grad_sd(detached_sd(head))
    out = head(feats)
    proj = det_input('head256.proj', out.shape, -1, 1)
    (out * proj).sum().backward()
    ofe = [f.detach().clone().requires_grad_(True) for f in feats]
    oout = olawin.lawin_head(sd, ofe, training=False)
    (oout * proj).sum().backward()
    check('head256.out', oout, out)
    save = {'out': np_(out)}
    for i in range(4):
        check(f'head256.dfeat{i}', ofe[i].grad, feats[i].grad, 5e-5)
        save[f'dfeat{i}'] = np_(feats[i].grad)
    save.update(devs('head256.'))
    np.savez_compressed(os.path.join(HERE, 'lawin_e256_eval.npz'), **save)


def gen_segmentor():
    """MiT-B0 + LawinHead(128, 6) + bilinear x4, eval, 2x3x128x128: fp32 and bf16-autocast logits."""
    enc = det_fill_(SegFormer('B0')).eval()
    head = det_fill_(LawinHead(enc.channels, 128, 6)).eval()
    x = det_input('seg.x', (2, 3, 128, 128))
    lab = det_labels('seg.lab', (2, 128, 128), 6, cell=8)

    def fwd():
        return F.interpolate(head(enc(x)), size=x.shape[-2:], mode='bilinear', align_corners=False)

    out = fwd()
    crit = OhemCrossEntropyLoss(0, 0.7, 0.1)
    loss = crit(out, lab)
    loss.backward()
    with torch.no_grad(), torch.autocast('cpu', dtype=torch.bfloat16):
        out_bf16 = fwd().float()
    oout = olawin.segmentor(detached_sd(enc), detached_sd(head), x, 'B0')
    check('seg.logits', oout, out)
    print('  bf16-autocast vs fp32 rel-L2', float((out_bf16 - out).norm() / out.norm()))
    save = {'logits': np_(out), 'logits_bf16': np_(out_bf16), 'loss': np_(loss),
            'grad.enc.block1.0.attn.q.weight': np_(enc.block1[0].attn.q.weight.grad),
            'grad.enc.patch_embed1.proj.weight': np_(enc.patch_embed1.proj.weight.grad),
            'grad.head.linear_pred.weight': np_(head.linear_pred.weight.grad),
            'grad.head.lawin_8.theta.weight': np_(head.lawin_8.theta.weight.grad)}
    np.savez_compressed(os.path.join(HERE, 'segmentor_b0_e128_128.npz'), **save)


def gen_losses():
    save = {}
    logits = (det_input('loss.logits', (2, 6, 32, 32), -3, 3))
    lab = det_labels('loss.lab', (2, 32, 32), 6, cell=4)
    lab_many_ignored = lab.clone()
    lab_many_ignored[:, :24] = 0
    cases = {'plain': (lab, 0.7, 0.1), 'ignored': (lab_many_ignored, 0.7, 0.1), 'nokeep': (lab, 0.7, 0.0),
             'half': (lab, 0.3, 0.5), 'noignore': (lab, 0.7, 0.25, 255)}
    for name, cfg in cases.items():
        tgt, thr, mk = cfg[:3]
        ign = cfg[3] if len(cfg) > 3 else 0
        ref = OhemCrossEntropyLoss(ign, thr, mk)
        ora = olosses.OhemCE(ign, thr, mk)
        for call in range(2):                     # two calls pin the stateful threshold
            lg = (logits * (1.0 + call)).clone().requires_grad_(True)
            l_ref = ref(lg, tgt)
            l_ref.backward()
            lg2 = (logits * (1.0 + call)).clone().requires_grad_(True)
            l_ora = ora(lg2, tgt)
            l_ora.backward()
            check(f'ohem.{name}.{call}.loss', l_ora, l_ref, 1e-6)
            check(f'ohem.{name}.{call}.grad', lg2.grad, lg.grad, 1e-5)
            assert float(ref.threshold) == float(ora.threshold)
            save[f'ohem.{name}.{call}.loss'] = np_(l_ref)
            save[f'ohem.{name}.{call}.grad'] = np_(lg.grad)
            save[f'ohem.{name}.{call}.thr'] = np.float32(float(ref.threshold))
    save['lab'], save['lab_many_ignored'], save['logits'] = np_(lab), np_(lab_many_ignored), np_(logits)
    probs = logits.softmax(1)
    for weighted in (False, True):
        for delta in (0.5, 0.3):
            p = probs.clone().requires_grad_(True)
            l_ref = DiceLoss(delta, weighted, 6)(p, lab)
            l_ref.backward()
            p2 = probs.clone().requires_grad_(True)
            l_ora = olosses.tversky_dice(p2, lab, 6, delta, weighted)
            l_ora.backward()
            check(f'dice.{weighted}.{delta}', l_ora, l_ref, 1e-6)
            check(f'dice.{weighted}.{delta}.grad', p2.grad, p.grad, 1e-5)
            save[f'dice.{int(weighted)}.{delta}.loss'] = np_(l_ref)
            save[f'dice.{int(weighted)}.{delta}.grad'] = np_(p.grad)
    for smooth in (0.0, 1.0):
        p = probs.clone().requires_grad_(True)
        l_ref = dice_loss(p, lab, smooth)
        l_ref.backward()
        p2 = probs.clone().requires_grad_(True)
        l_ora = olosses.dice_functional(p2, lab, smooth)
        l_ora.backward()
        check(f'dicef.{smooth}', l_ora, l_ref, 1e-6)
        check(f'dicef.{smooth}.grad', p2.grad, p.grad, 1e-5)
        save[f'dicef.{smooth}.loss'] = np_(l_ref)
        save[f'dicef.{smooth}.grad'] = np_(p.grad)
    w_ref = class_weights(lab_many_ignored, torch.arange(8))
    check('class_weights', olosses.class_weights(lab_many_ignored, torch.arange(8)), w_ref, 1e-6)
    save['class_weights8'] = np_(w_ref)
    np.savez_compressed(os.path.join(HERE, 'losses.npz'), **save)


def gen_convnext():
    from modules.convnext import ConvNeXt  # reference
    enc = det_fill_(ConvNeXt('X')).eval()
    x = det_input('cnx.x', (2, 3, 64, 64))
    outs = enc(x)
    projs = [det_input(f'cnx.proj{i}', o.shape, -1, 1) for i, o in enumerate(outs)]
    sum((o * p).sum() for o, p in zip(outs, projs)).backward()
    grads = {k: p.grad for k, p in enc.named_parameters()}
    sd = grad_sd(detached_sd(enc))
    oouts = oconvnext.convnext_forward(sd, x, 'X')
    sum((o * p).sum() for o, p in zip(oouts, projs)).backward()
    keys = ['downsample_layers.0.0.weight', 'downsample_layers.0.1.weight', 'downsample_layers.2.0.bias', 'downsample_layers.2.1.weight',
            'stages.0.0.dw_conv.weight', 'stages.0.0.dw_conv.bias', 'stages.0.1.norm.weight', 'stages.1.0.pw_conv1.weight',
            'stages.1.1.pw_conv2.weight', 'stages.1.1.pw_conv2.bias', 'stages.2.3.gamma', 'stages.0.0.gamma', 'stages.3.1.dw_conv.weight',
            'norm0.weight', 'norm3.bias']
    save = {}
    for i in range(4):
        check(f'cnx.out{i}', oouts[i], outs[i])
        save[f'out{i}'] = np_(outs[i])
    for k in keys:
        check(f'cnx.grad.{k}', sd[k].grad, grads[k], 5e-5)
        save['grad.' + k] = np_(grads[k])
    enc64 = det_fill_(ConvNeXt('X')).double().eval()
    outs64 = enc64(x.double())
    sum((o * p.double()).sum() for o, p in zip(outs64, projs)).backward()
    g64 = {k: p.grad for k, p in enc64.named_parameters()}
    for i in range(4):
        f64dev(save, f'out{i}', outs[i], outs64[i])
    for k in keys:
        f64dev(save, 'grad.' + k, grads[k], g64[k])
    enc_b = det_fill_(ConvNeXt('X')).eval()
    with torch.autocast('cpu', dtype=torch.bfloat16):
        outs_b = enc_b(x)
    sum((o.float() * p).sum() for o, p in zip(outs_b, projs)).backward()
    gb = {k: p.grad for k, p in enc_b.named_parameters()}
    for k in keys:
        save['bf16.grad.' + k] = np_(gb[k])
    save.update(devs('cnx.'))
    np.savez_compressed(os.path.join(HERE, 'convnext_x_64.npz'), **save)


def gen_dnet():
    """DNet(3, 6, embed_dim=128, stem 'X') in TRAINING mode (batch-stat BatchNorms, Dropout2d p=0), 2x3x128x128: probabilities,
    Tversky-Dice loss and a few gradients; running buffers after the step."""
    from core.dnet import DNet  # reference
    net = det_fill_(DNet(3, 6, embed_dim=128, stem_code='X')).train()
    net.decoder.decoder.dropout.p = 0.0
    sd0 = detached_sd(net)
    x = det_input('dnet.x', (2, 3, 128, 128))
    lab = det_labels('dnet.lab', (2, 128, 128), 6, cell=8)
    probs = net(x)
    loss = DiceLoss(0.5, False, 6)(probs, lab)
    loss.backward()
    grads = {k: p.grad for k, p in net.named_parameters()}
    sd = grad_sd(sd0)
    oprobs = oconvnext.dnet_forward(sd, x, 'X', training=True)
    oloss = olosses.tversky_dice(oprobs, lab, 6, 0.5, False)
    oloss.backward()
    check('dnet.probs', oprobs, probs)
    check('dnet.loss', oloss, loss, 1e-6)
    keys = ['terminator.terminator.0.weight', 'terminator.terminator.0.bias', 'decoder.deconvolution.net.0.weight',
            'decoder.deconvolution.net.1.weight', 'decoder.deconvolution.net.1.bias', 'decoder.feature_reducer.net.p_conv.conv.weight',
            'decoder.feature_reducer.net.p_conv.conv.bias', 'decoder.decoder.linear_pred.weight', 'decoder.decoder.cat.conv.weight',
            'encoder.encoder.stages.3.1.pw_conv2.weight', 'encoder.encoder.stages.0.0.dw_conv.weight', 'encoder.encoder.stages.1.0.gamma',
            'encoder.encoder.downsample_layers.0.0.weight']
    save = {'probs': np_(probs), 'loss': np_(loss)}
    for k in keys:
        check(f'dnet.grad.{k}', sd[k].grad, grads[k], 1e-4)
        save['grad.' + k] = np_(grads[k])
    after = net.state_dict()
    for k in ('decoder.deconvolution.net.1.running_mean', 'decoder.deconvolution.net.1.running_var', 'decoder.deconvolution.net.1.num_batches_tracked'):
        check(f'dnet.after.{k}', sd[k].float(), after[k].float())
        save['after.' + k] = np_(after[k])
    net64 = det_fill_(DNet(3, 6, embed_dim=128, stem_code='X')).double().train()
    net64.decoder.decoder.dropout.p = 0.0
    p64 = net64(x.double())
    DiceLoss(0.5, False, 6)(p64, lab).backward()
    g64 = {k: p.grad for k, p in net64.named_parameters()}
    f64dev(save, 'probs', probs, p64)
    for k in keys:
        f64dev(save, 'grad.' + k, grads[k], g64[k])
    # the reference's OWN bf16 path (torch.autocast) on the same inputs, as gen_head does: the noise floor of a bf16 run
    net_b = det_fill_(DNet(3, 6, embed_dim=128, stem_code='X')).train()
    net_b.decoder.decoder.dropout.p = 0.0
    with torch.autocast('cpu', dtype=torch.bfloat16):
        pb = net_b(x)
    lb = DiceLoss(0.5, False, 6)(pb.float(), lab)
    lb.backward()
    gb = {k: p.grad for k, p in net_b.named_parameters()}
    save['bf16.probs'], save['bf16.loss'] = np_(pb.float()), np_(lb)
    for k in keys:
        save['bf16.grad.' + k] = np_(gb[k])
    print('  bf16-autocast vs fp32: probs rel-L2', float((pb.float() - probs).norm() / probs.norm()),
          ' worst grad rel-L2', max(float((gb[k] - grads[k]).norm() / grads[k].norm()) for k in keys))
    save.update(devs('dnet.'))
    np.savez_compressed(os.path.join(HERE, 'dnet_x_128.npz'), **save)


if __name__ == '__main__':
    only = sys.argv[1:]
    gens = {'mit': gen_mit, 'head_eval': lambda: gen_head(False), 'head_train': lambda: gen_head(True), 'head_e256': gen_head_e256, 'segmentor': gen_segmentor,
            'losses': gen_losses, 'convnext': gen_convnext, 'dnet': gen_dnet}
    for fn in [g for n, g in gens.items() if not only or n in only]:
        fn()
    print('golden vectors written to', HERE)
