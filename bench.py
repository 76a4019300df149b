"""Headline benchmark: train images/sec of MiT-B2 + Lawin(512) at 1024x1024, bf16, per-GPU batch 16 (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = zero grads -> forward (bf16 activations, fp32 parameters/statistics) -> OhemCrossEntropyLoss(0, 0.7, 0.1) ->
SegmentationMetrics.update (confusion matrix, as core/igniter.py:104 does every training step) -> backward ->
bucketed gradient all-reduce (RCCL, N > 1) -> AdaBound step.  Inputs are synthetic UAVPal-shaped tiles resident in HBM.
Rank 0 prints ONE JSON line; `roofline` describes the kernel family that dominates the step, measured live with HIP events
on the launch stream in one extra instrumented step; `cpu_baseline` times the CPU oracle (port of the reference path) on
two images of the same workload (same model, tile size and loss, fp32) on the host cores (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA (no sparsity)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='mit_lawin', choices=['mit_lawin', 'dnet', 'topformer'],
                    help="mit_lawin = BASELINE configs[1] (the headline); dnet = configs[2] per-GPU step (DNet-T, embed 256, batch 8, Dice loss)")
    ap.add_argument('--variant', default='B2')
    ap.add_argument('--head', default='simple', choices=['simple', 'refiner'], help='TopFormer head (configs[3]; predict.py uses refiner)')
    ap.add_argument('--output', default='logits', choices=['logits', 'class_map', 'class_map_unfused'],
                    help='topformer: what a step hands over -- full-size logits (the model forward), the uint8 class map of '
                         'predict.py (fused resize + arg-max), or forward followed by the separate class-map kernel')
    ap.add_argument('--loss', default='ohem_ce', choices=['ohem_ce', 'ce_dice'], help='mit_lawin: OHEM-CE (configs[1]) or OHEM-CE + Dice on the softmax (configs[4])')
    ap.add_argument('--embed', type=int, default=512)
    ap.add_argument('--batch', type=int, default=16, help='per-GPU batch')
    ap.add_argument('--size', type=int, default=1024)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--table', action='store_true', help='print the per-entry-point time table to stderr')
    ap.add_argument('--shapes', action='store_true', help='print per-shape GEMM / attention timings to stderr')
    ap.add_argument('--train', action='store_true', help='topformer: the training step of train.py:45-52 (TopFormer-B, fwd + OHEM-CE + confusion '
                    'matrix + bwd + grad all-reduce + AdaBound; default 1024x1024, batch 16) instead of configs[3] inference')
    ap.add_argument('--force-comm', action='store_true', help='N = 1: create the RCCL process group (world size 1) and run the bucketed '
                    'all-reduce path (side stream, hooks, collectives) anyway, so the real backend is exercised on one GPU')
    return ap.parse_args()


def synthetic_batch(batch, size, rank, device):
    g = torch.Generator().manual_seed(1234 + rank)
    image = torch.rand(batch, 3, size, size, generator=g)
    small = torch.randint(0, 6, (batch, size // 32, size // 32), generator=g)
    label = small.repeat_interleave(32, 1).repeat_interleave(32, 2).contiguous()
    return image.to(device), label.to(device)


# ---------------------------------------------------------------- algorithmic work of each C-ABI entry point
def _sz(dtype_code):
    return 4 if dtype_code == 0 else 2


def _conv_patch_dims(name, a):
    B, H, W, C, s, N = {'ss_conv_patch_fwd': a[2:7] + (a[12],), 'ss_conv_patch_dgrad': a[6:12], 'ss_conv_patch_wgrad': a[4:10]}[name]
    return B * (H // s) * (W // s), N, s * s * C           # patches, output channels, patch length


def entry_work(name, a):
    """-> (bound, algorithmic flops or bytes) for one launch, from its argument list (include/semseg_hip.h order)."""
    if name.startswith('ss_conv_patch_'):
        M, N, K = _conv_patch_dims(name, a)
        return 'mfma', 2.0 * M * N * K
    if name == 'ss_gemm_nt':
        return 'mfma', 2.0 * a[8] * a[9] * a[10]
    if name == 'ss_gemm_nt_bn_stats':
        return 'mfma', 2.0 * a[7] * a[8] * a[9]
    if name in ('ss_gemm_tn', 'ss_gemm_tn_colsum', 'ss_gemm_tn_colsum_keep'):
        return 'mfma', 2.0 * a[7] * a[8] * a[9]
    if name == 'ss_mit_attn_fwd':
        B, N, Nk, h, D = a[5:10]
        return 'mfma', 4.0 * B * h * N * Nk * D
    if name == 'ss_mit_attn_bwd':
        B, N, Nk, h, D = a[8:13]
        return 'mfma', 10.0 * B * h * N * Nk * D
    if name == 'ss_layernorm_fwd':
        return 'hbm', 2.0 * a[7] * a[8] * _sz(a[0])
    if name == 'ss_layernorm_bwd':
        return 'hbm', (3.0 + (1.0 if a[7] else 0.0) + (1.0 if a[2] else 0.0)) * a[11] * a[12] * _sz(a[0])
    if name == 'ss_dwconv3_fwd':
        return 'hbm', 2.0 * a[5] * a[6] * a[7] * a[8] * _sz(a[0])
    if name == 'ss_dwconv3_bwd':
        return 'hbm', 3.0 * a[7] * a[8] * a[9] * a[10] * _sz(a[0])
    if name == 'ss_bn_stats':
        return 'hbm', 1.0 * a[3] * a[4] * _sz(a[0])
    if name == 'ss_bn_apply':
        return 'hbm', (2.0 + (1.0 if a[6] else 0.0)) * a[12] * a[13] * _sz(a[0])
    if name == 'ss_bn_bwd':
        return 'hbm', 5.0 * a[14] * a[15] * _sz(a[0])
    # BatchNorm of a x2 bilinear resize without the resized tensor: the coarse map in (twice for the two backward passes), the
    # fine map out / the fine gradient in twice; the classifier-fused tail: z in (twice), logits / logit gradients, dz out
    if name == 'ss_bn_stats_up':
        B, hi, wi, ho, wo, C = a[4:10]
        return 'hbm', 1.0 * B * hi * wi * C * _sz(a[0])
    if name == 'ss_bn_apply_up':
        B, hi, wi, ho, wo = a[4:9]
        return 'hbm', 1.0 * B * (hi * wi + ho * wo) * a[14] * _sz(a[0])
    if name == 'ss_bn_bwd_up':
        B, hi, wi, ho, wo = a[6:11]
        return 'hbm', 1.0 * B * (2 * hi * wi + 3 * ho * wo) * a[18] * _sz(a[0])
    if name == 'ss_bn_bwd_up2_dx':
        B, hi, wi = a[6:9]
        return 'hbm', 1.0 * B * (3 * hi * wi + 2 * 4 * hi * wi) * a[16] * _sz(a[0])
    if name == 'ss_bn_bwd_coef':
        return 'hbm', 3.0 * a[11] * a[12] * _sz(a[0])
    if name == 'ss_bn_apply_classify':
        return 'hbm', 1.0 * a[13] * (a[14] + a[10]) * _sz(a[0])
    if name == 'ss_bn_bwd_classify':
        return 'hbm', 1.0 * a[17] * (3 * a[18] + 2 * a[12]) * _sz(a[0])
    if name == 'ss_colsum':
        return 'hbm', 1.0 * a[3] * a[4] * a[5] * _sz(a[0])
    if name == 'ss_add':
        return 'hbm', 3.0 * a[4] * _sz(a[0])
    if name in ('ss_bilinear_fwd', 'ss_bilinear_fwd_add'):
        B, hi, wi, ho, wo, C = a[5:11]
        return 'hbm', 1.0 * B * C * (hi * wi + ho * wo) * _sz(a[0])
    if name == 'ss_bilinear_fwd2':       # (dtype, x, ldx, hi, wi, x2, ldx2, hi2, wi2, y, ldy, B, ho, wo, C)
        return 'hbm', 1.0 * a[11] * a[14] * (a[3] * a[4] + a[7] * a[8] + a[12] * a[13]) * _sz(a[0])
    if name == 'ss_bilinear_bwd':
        B, hi, wi, ho, wo, C = a[5:11]
        return 'hbm', 1.0 * B * C * (hi * wi + ho * wo) * _sz(a[0])
    if name == 'ss_im2col':
        B, H, W, C, k, s, pad, Kp = a[3:11]
        ho, wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        return 'hbm', 1.0 * (B * H * W * C + B * ho * wo * Kp) * _sz(a[0])
    if name == 'ss_col2im':
        B, H, W, C, k, s, pad = a[3:10]
        ho, wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        return 'hbm', 1.0 * (B * H * W * C + B * ho * wo * k * k * C) * _sz(a[0])
    if name in ('ss_ohem_ce_fwd', 'ss_confusion_update'):
        B, C, H, W = (a[7:11] if name == 'ss_ohem_ce_fwd' else a[8:12])
        return 'hbm', 1.0 * B * H * W * (C * _sz(a[0]) + 8)
    if name == 'ss_ohem_ce_bwd':
        B, C, H, W = a[15:19]
        return 'hbm', 1.0 * B * H * W * (2 * C * _sz(a[0]) + 8 + 4)
    if name == 'ss_adabound_step':
        return 'hbm', 28.0 * a[4]
    if name == 'ss_dwconv3_bwd_fused':
        return 'hbm', 3.0 * a[7] * a[8] * a[9] * a[10] * _sz(a[0])
    if name == 'ss_dwconv3_bwd_fc2':     # (dtype, dout, w2t, keep, x, w9c, bias, dx, dw_db, B, H, W, C, K, ...): x in, dx out, dout in; da2 stays on chip
        return 'hbm', 1.0 * a[9] * a[10] * a[11] * (2 * a[12] + a[13]) * _sz(a[0])
    if name == 'ss_mixffn_fwd':          # (dtype, x, w1, b1, w9, bd, w2, b2, res, keep, out, h1_out, a2_out, B, H, W, C, Ch): x (+ res) in, out, and the two
        B, H, W, C, Ch = a[13:18]        # hidden tensors out when the backward will read them; the products cost a tenth of that on the matrix cores
        return 'hbm', 1.0 * B * H * W * ((2 + (1 if a[8] else 0)) * C + (2 * Ch if a[11] else 0)) * _sz(a[0])
    if name == 'ss_rowscale':
        return 'hbm', 2.0 * a[5] * a[6] * _sz(a[0])
    if name in ('ss_lawin_attn_fwd', 'ss_lawin_attn_bwd'):          # window attention: HBM-bound at these head dims (64 keys per window)
        B, H, W, inter = (a[7:11] if name.endswith('fwd') else a[12:16])
        rows = B * H * W
        return 'hbm', (4.0 if name.endswith('fwd') else 8.0) * rows * inter * _sz(a[0])   # theta, phi|g, y  (+ dy, dtheta, dphi|dg)
    if name == 'ss_lawin_mix':
        B, H, W, C = a[6:10]
        return 'hbm', 2.0 * B * H * W * C * _sz(a[0])
    if name == 'ss_lawin_mix_wgrad':
        B, H, W, C = a[4:8]
        return 'hbm', 2.0 * B * H * W * C * _sz(a[0])
    return 'hbm', None


def entry_bytes(name, a):
    """algorithmic HBM bytes (every operand touched once) of the matrix-core entry points, for the per-launch roofline max(flops, bytes)."""
    if name == 'ss_gemm_nt':
        M, N, K = a[8:11]
        return (M * K + N * K + M * N * (2 if a[12] else 1)) * _sz(a[0])
    if name == 'ss_gemm_nt_bn_stats':
        M, N, K = a[7:10]
        return (M * K + N * K + M * N * (2 if a[10] else 1)) * _sz(a[0])
    if name in ('ss_gemm_tn', 'ss_gemm_tn_colsum', 'ss_gemm_tn_colsum_keep'):
        M, N, K = a[7:10]
        return (K * M + K * N) * _sz(a[0]) + M * N * 4
    if name.startswith('ss_conv_patch_'):
        M, N, K = _conv_patch_dims(name, a)
        return (M * K + M * N) * _sz(a[0]) + N * K * (4 if name.endswith('wgrad') else _sz(a[0]))
    if name == 'ss_mit_attn_fwd':
        B, N, Nk, h, D = a[5:10]
        return (2.0 * B * N * h * D + 2.0 * B * Nk * h * D) * _sz(a[0])            # q, out; k, v
    if name == 'ss_mit_attn_bwd':
        B, N, Nk, h, D = a[8:13]
        return (4.0 * B * N * h * D + 4.0 * B * Nk * h * D) * _sz(a[0])            # q, o, do, dq; kv, dkv
    return None


def kernel_table(trace):
    torch.cuda.synchronize()
    tab = {}
    for name, args, e0, e1 in trace:
        ms = e0.elapsed_time(e1)
        bound, work = entry_work(name, args)
        t = tab.setdefault(name, {'calls': 0, 'ms': 0.0, 'work': 0.0, 'bound': bound, 'known': True, 'bytes': 0.0, 'ideal_ms': 0.0})
        t['calls'] += 1
        t['ms'] += ms
        if work is not None:
            # roofline time of THIS launch: max(flops / MFMA peak, algorithmic bytes / HBM peak) -- most GEMM launches are projections
            # with K, N of a few hundred, bounded by HBM rather than by the matrix cores
            byt = entry_bytes(name, args) if bound == 'mfma' else work
            t['bytes'] += byt or 0.0
            t['ideal_ms'] += 1e3 * max(work / (MFMA_BF16_PEAK_TFLOPS * 1e12) if bound == 'mfma' else 0.0, (byt or 0.0) / (HBM_PEAK_GBS * 1e9))
        if work is None:
            t['known'] = False
        else:
            t['work'] += work
    return tab


def shape_table(trace):
    """per-shape timings of the GEMM entry points: TFLOP/s and the HBM rate implied by touching each operand once."""
    rows = {}
    for name, a, e0, e1 in trace:
        if name == 'ss_gemm_nt':
            M, N, K = a[8:11]
            sz = _sz(a[0])
            key = (name, M, N, K, 'res' if a[12] else '', 'f32out' if a[1] else '')
            byt = (M * K + N * K + M * N * (2 if a[12] else 1)) * sz
        elif name == 'ss_gemm_nt_bn_stats':
            M, N, K = a[7:10]
            key = ('ss_gemm_nt', M, N, K, 'res' if a[10] else '', 'stats')
            byt = (M * K + N * K + M * N * (2 if a[10] else 1)) * _sz(a[0])
        elif name in ('ss_gemm_tn', 'ss_gemm_tn_colsum', 'ss_gemm_tn_colsum_keep'):
            M, N, K = a[7:10]
            key = ('ss_gemm_tn', M, N, K, 'cs' if name.endswith('colsum') else '', '')
            byt = (K * M + K * N) * _sz(a[0]) + M * N * 4
        elif name.startswith(('ss_lawin', 'ss_mit_attn', 'ss_bn_', 'ss_dwconv', 'ss_colsum', 'ss_bilinear')):
            key = (name, 0, 0, 0, ' '.join(str(v) for v in a[1:] if isinstance(v, int) and 0 <= v < (1 << 22)), '')
            M = N = K = 0
            _, w = entry_work(name, a)
            byt = w or 0
        else:
            continue
        r = rows.setdefault(key, [0, 0.0, 2.0 * M * N * K, byt])
        r[0] += 1
        r[1] += e0.elapsed_time(e1)
    for key, (n, ms, fl, byt) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        per = ms / n
        if key[1] == 0:
            print(f"{key[0]:22s} [{key[4]}] x{n:3d} {ms:8.2f} ms {per * 1e3:8.1f} us/call work-rate {byt / per / 1e6:9.0f} M/s", file=sys.stderr)
            continue
        print(f"{key[0]:11s} M={key[1]:8d} N={key[2]:5d} K={key[3]:8d} {key[4]:3s} {key[5]:6s} x{n:3d} {ms:8.2f} ms  {per * 1e3:8.1f} us/call "
              f"{fl / per / 1e9:7.1f} TFLOP/s {byt / per / 1e6:7.0f} GB/s", file=sys.stderr)


FAMILIES = (('gemm', ('ss_gemm', 'ss_conv_patch')), ('mit_attention', ('ss_mit_attn',)), ('depthwise_conv', ('ss_dwconv',)), ('batchnorm', ('ss_bn_',)),
            ('layernorm', ('ss_layernorm',)), ('lawin_window_attention', ('ss_lawin',)), ('resample', ('ss_bilinear', 'ss_im2col', 'ss_col2im')),
            ('loss_metrics', ('ss_ohem', 'ss_dice', 'ss_confusion', 'ss_label_hist')), ('optimizer', ('ss_adabound',)))


def family_of(name):
    for fam, prefixes in FAMILIES:
        if name.startswith(prefixes):
            return fam
    return 'other'


def family_table(tab):
    fam = {}
    for name, t in tab.items():
        f = fam.setdefault(family_of(name), {'calls': 0, 'ms': 0.0, 'work': 0.0, 'bound': t['bound'], 'known_ms': 0.0, 'bytes': 0.0, 'ideal_ms': 0.0})
        for k in ('calls', 'ms', 'bytes', 'ideal_ms'):
            f[k] += t.get(k, 0.0)
        if t['known']:
            f['work'] += t['work']
            f['known_ms'] += t['ms']
        if t['bound'] == 'mfma':
            f['bound'] = 'mfma'
    return fam


def roofline_from_table(tab, traffic_ok=False):
    fam = family_table(tab)
    total = sum(t['ms'] for t in tab.values())
    name, f = max(fam.items(), key=lambda kv: kv[1]['ms'])
    out = {'kernel': 'gemm_kernel (ss_gemm_nt + ss_gemm_tn + ss_conv_patch_*)' if name == 'gemm' else name, 'bound': f['bound'],
           'launches_per_step': f['calls'], 'avg_launch_us': 1e3 * f['ms'] / f['calls'], 'step_share': f['ms'] / total}
    if f['known_ms'] > 0:
        per_s = f['work'] / (f['known_ms'] * 1e-3)
        if f['bound'] == 'mfma':
            out.update(achieved=per_s / 1e12, peak=MFMA_BF16_PEAK_TFLOPS, unit='TFLOP/s')
        else:
            out.update(achieved=per_s / 1e9, peak=HBM_PEAK_GBS, unit='GB/s')
        out['frac'] = out['achieved'] / out['peak']
    else:
        out.update(achieved=None, peak=None, unit=None, frac=None)
    out['traffic'] = None
    if f['ideal_ms'] > 0:
        # per-launch roofline: sum over launches of max(flops / MFMA peak, algorithmic bytes / HBM peak), over the measured time
        out['algorithmic_bytes_per_launch'] = f['bytes'] / f['calls']
        out['frac_of_per_shape_roofline'] = f['ideal_ms'] / f['ms']
    if name == 'gemm':
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r03_pmc_hbm_traffic.json')
        if traffic_ok and os.path.exists(pmc):
            with open(pmc) as fh:
                g = json.load(fh)['gemm_family']
            out['traffic'] = g['hbm_bytes_per_launch']
            out['traffic_source'] = 'profiles/r03_pmc_hbm_traffic.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE passes of this command, tools/profile_pmc.sh)'
    # the five families with the largest share, each against its own bound: `frac` = achieved / peak of the bound it is priced on,
    # `frac_of_per_launch_roofline` = sum of the launches' max(flops / MFMA peak, bytes / HBM peak) over the measured time
    top = []
    for fname, ff in sorted(fam.items(), key=lambda kv: -kv[1]['ms'])[:5]:
        e = {'family': fname, 'ms': ff['ms'], 'share': ff['ms'] / total, 'launches': ff['calls'], 'bound': ff['bound']}
        if ff['known_ms'] > 0:
            per_s = ff['work'] / (ff['known_ms'] * 1e-3)
            e['frac'] = per_s / (MFMA_BF16_PEAK_TFLOPS * 1e12 if ff['bound'] == 'mfma' else HBM_PEAK_GBS * 1e9)
            e['frac_of_per_launch_roofline'] = ff['ideal_ms'] / ff['known_ms'] if ff['ideal_ms'] > 0 else None
        top.append(e)
    out['families'] = top
    # what the step would take if every launch with a known work formula ran at its roofline (the rest is carried at its measured time)
    known_ms = sum(t['ms'] for t in tab.values() if t['known'])
    out['model_roofline_ms'] = sum(t['ideal_ms'] for t in tab.values() if t['known']) + (total - known_ms)
    out['model_roofline_coverage'] = known_ms / total
    return out


# ---------------------------------------------------------------- CPU baseline: the oracle (port of the reference path) on configs[0]
def cpu_baseline():
    """The oracle (CPU restatement) on a bounded sample of the headline workload: the same model (MiT-B2 + Lawin 512), tile size
    (1024 x 1024), loss and fwd + bwd, two images of the batch instead of 16 (the head's image-pool BatchNorm needs two in training mode), fp32, two steps."""
    from oracle import lawin as olawin, losses as olosses
    from semseg_amd.detinit import det_state_dict
    from semseg_amd import shapes
    cores = min(os.cpu_count() or 1, 16)                        # the box's CPU share for one GPU
    torch.set_num_threads(cores)
    sd_e = det_state_dict(shapes.mit_shapes('B2'))
    sd_h = det_state_dict(shapes.lawin_shapes([64, 128, 320, 512], 512, 6))
    for sd in (sd_e, sd_h):
        for k, v in sd.items():
            if v.is_floating_point() and 'running' not in k:
                v.requires_grad_(True)
    x, lab = synthetic_batch(2, 1024, 0, 'cpu')
    crit = olosses.OhemCE(0, 0.7, 0.1)
    times = []
    for _ in range(2):
        t0 = time.time()
        out = olawin.segmentor(sd_e, sd_h, x, 'B2', training=True)
        loss = crit(out, lab)
        loss.backward()
        for sd in (sd_e, sd_h):
            for v in sd.values():
                v.grad = None
        times.append(time.time() - t0)
    t = min(times)
    return {'value': 2.0 / t, 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'sample': 'the headline workload (MiT-B2 + Lawin(512, 6 cls), 1024x1024, fwd + OHEM-CE + bwd) on 2 images of the batch instead of 16, '
                      'fp32, faster of two steps (%.1f s, %.1f s)' % tuple(times)}


def topformer_inference(a, world, rank, dev):
    """configs[3]: TopFormer-B (SimpleHead or RefinerHead) inference images/sec on 2048x2048 tiles, replicas only (no collective on the data path)."""
    from semseg_amd.topformer import TopFormerModel
    from semseg_amd.detinit import det_fill_
    from semseg_amd.runtime import set_compute_dtype
    from semseg_amd import _lib
    set_compute_dtype(torch.bfloat16 if a.dtype == 'bf16' else torch.float32)
    size = a.size if a.size != 1024 else 2048
    batch = a.batch if a.batch != 16 else 4
    net = det_fill_(TopFormerModel(6, 'B', head_cfg=frozenset({'alias': a.head}.items()))).to(dev).eval()
    x, _ = synthetic_batch(batch, size, rank, dev)
    from semseg_amd import postproc
    step = {'logits': net, 'class_map': net.predict_class_map, 'class_map_unfused': lambda t: postproc.class_map(net(t))}[a.output]

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        step(x)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step(x)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    _lib.TRACE = []
    step(x)
    trace, _lib.TRACE = _lib.TRACE, None
    if rank == 0:
        table = kernel_table(trace)
        res = {'metric': f'inference images/sec @{size}x{size} (TopFormer-B)', 'value': batch * world * a.steps / dt, 'unit': 'images/s', 'n_gpus': world,
               'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * dt / a.steps, 'higher_is_better': True, 'scaling': 'weak',
               'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
               'config': {'workload': f'configs[3]: TopFormer-B + {a.head} head inference, {size}x{size}, output {a.output}', 'per_gpu_batch': batch,
                          'global_batch': batch * world, 'parallelism': f'replicas x{world}'},
               'roofline': roofline_from_table(table)}
        if a.table:
            tot = sum(t['ms'] for t in table.values())
            for name, t in sorted(table.items(), key=lambda kv: -kv[1]['ms']):
                print(f"{name:28s} {t['calls']:5d} calls {t['ms']:9.2f} ms {100 * t['ms'] / tot:5.1f}%", file=sys.stderr)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


def main():
    a = parse()
    if os.environ.get('SEMSEG_BENCH_DEBUG'):                      # dump every thread's stack if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['SEMSEG_BENCH_DEBUG']), exit=True)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        raise SystemExit(f'bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch N > 1 through torch.distributed.run')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback)')
    local = local % torch.cuda.device_count()       # rehearsal of N ranks on fewer GPUs (SEMSEG_BENCH_BACKEND=gloo); one rank per GPU otherwise
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    comm_on = world > 1 or a.force_comm
    if comm_on:
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1'), os.environ.setdefault('MASTER_PORT', '29533')
            os.environ.setdefault('RANK', '0'), os.environ.setdefault('WORLD_SIZE', '1')
        backend = os.environ.get('SEMSEG_BENCH_BACKEND', 'nccl')            # nccl == RCCL over xGMI on ROCm
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from semseg_amd.model import SegFormerLawin
    from semseg_amd.loss import OhemCrossEntropyLoss
    from semseg_amd.metrics import SegmentationMetrics
    from semseg_amd.detinit import det_fill_
    from semseg_amd.runtime import set_compute_dtype
    from semseg_amd.engine import FlatParams, GradReducer, AdaBound, broadcast_module_state
    from semseg_amd import _lib

    T = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
    set_compute_dtype(T)
    if a.workload == 'topformer' and not a.train:
        return topformer_inference(a, world, rank, dev)
    if a.workload == 'topformer':
        from semseg_amd.topformer import TopFormerModel
        net = det_fill_(TopFormerModel(6, 'B', head_cfg=frozenset({'alias': a.head}.items())))
    elif a.workload == 'dnet':
        from semseg_amd.decoder import DNet
        from semseg_amd.loss import DiceLoss
        net = det_fill_(DNet(3, 6, embed_dim=256, stem_code='T'))
    else:
        net = SegFormerLawin(a.variant, a.embed, 6)
        det_fill_(net.encoder), det_fill_(net.decoder)
    net = net.to(dev).train()
    broadcast_module_state(net)
    flat = FlatParams(net)
    reducer = GradReducer(flat, force_comm=a.force_comm)
    opt = AdaBound(flat, lr=1e-3, final_lr=0.1)
    if a.workload == 'dnet':
        crit = DiceLoss(0.5, False, 6)
    elif a.loss == 'ce_dice':
        from semseg_amd.loss import OhemCEDiceLoss
        crit = OhemCEDiceLoss(6, 0, 0.7, 0.1)
    else:
        crit = OhemCrossEntropyLoss(0, 0.7, 0.1)
    met = SegmentationMetrics(6, ignore_index=0).to(dev)
    x, lab = synthetic_batch(a.batch, a.size, rank, dev)

    def step():
        flat.zero_grad()
        out = net(x)
        loss = crit(out, lab)
        met.update(out, lab)
        loss.backward()
        opt.step(reducer.wait())
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    cm = met.compute()                                           # int64 confusion-matrix all-reduce (C2)
    roof, table, comm = None, None, None
    if comm_on:                                                  # one more step with the bucket all-reduces timed on their stream
        reducer.time_comm = True
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        step_ms = 1e3 * (time.perf_counter() - t1)
        reducer.time_comm = False
        c = torch.tensor([reducer.comm_ms(), step_ms], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(c, op=dist.ReduceOp.MAX)
        comm = {'world_size': dist.get_world_size(), 'backend': dist.get_backend(), 'allreduce_ms_per_step': float(c[0]),
                'allreduce_share_of_step': float(c[0] / c[1]), 'buckets': len(reducer.buckets),
                'note': 'sum of the bucket all-reduce durations on the communication stream (they overlap the backward), max over ranks, '
                        'over the wall time of that step'}
    if not a.no_roofline:                                        # every rank runs the extra step (it contains collectives)
        _lib.TRACE = []
        step()
        trace, _lib.TRACE = _lib.TRACE, None
        if rank == 0:
            table = kernel_table(trace)
            # the committed PMC traffic figure was collected on the default headline configuration only
            default_cfg = a.loss == 'ohem_ce' and a.variant == 'B2' and a.embed == 512 and a.batch == 16 and a.size == 1024 and a.dtype == 'bf16'
            roof = roofline_from_table(table, traffic_ok=default_cfg)
            if a.shapes:
                shape_table(trace)
    fence()
    if rank != 0:
        if comm_on:
            dist.destroy_process_group()
        return
    ms = 1e3 * dt / a.steps
    what = {'dnet': 'DNet ConvNeXt-T', 'topformer': f'TopFormer-B + {a.head} head'}.get(a.workload, f'MiT-{a.variant}+Lawin')
    res = {'metric': f'train images/sec @{a.size}x{a.size} ({what})', 'value': a.batch * world * a.steps / dt, 'unit': 'images/s',
           'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'weak',
           'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
           'config': {'workload': (f'configs[2] per-GPU step: DNet(ConvNeXt-T + Decoder(embed 256), 6 cls) training step (fwd + Dice + confusion '
                                   f'matrix + bwd + grad all-reduce + AdaBound), {a.size}x{a.size}') if a.workload == 'dnet' else
                                  (f'train.py:45-52 model: TopFormer-B + {a.head} head (6 cls) training step (fwd + OHEM-CE + confusion matrix + bwd + '
                                   f'grad all-reduce + AdaBound), {a.size}x{a.size}') if a.workload == 'topformer' else
                                  f'configs[1]: SegFormer MiT-{a.variant} + LawinHead({a.embed}, 6 cls) training step '
                                  f'(fwd + {"OHEM-CE + Dice" if a.loss == "ce_dice" else "OHEM-CE"} + confusion matrix + bwd + grad all-reduce + AdaBound), {a.size}x{a.size}',
                      'per_gpu_batch': a.batch, 'global_batch': a.batch * world, 'parallelism': f'dp{world}',
                      'weights': 'deterministic random init'},
           'loss': float(loss.detach()), 'mIoU': float(cm['mIoU']), 'peak_mem_gib': torch.cuda.max_memory_allocated() / 2 ** 30}
    if roof is not None:
        res['roofline'] = roof
    if comm is not None:
        res['communicator'] = comm
    if table is not None:                                        # the rate with the optimizer step excluded, beside `value`
        opt_ms = sum(t['ms'] for name, t in table.items() if 'adabound' in name)
        res['optimizer_ms_per_step'] = opt_ms
        res['value_without_optimizer'] = a.batch * world / ((ms - opt_ms) * 1e-3)
    if table is not None and a.table:
        tot = sum(t['ms'] for t in table.values())
        for name, t in sorted(table.items(), key=lambda kv: -kv[1]['ms']):
            rate = ''
            if t['known'] and t['ms'] > 0:
                r = t['work'] / (t['ms'] * 1e-3)
                rate = f"{r / 1e12:8.1f} TFLOP/s" if t['bound'] == 'mfma' else f"{r / 1e9:8.0f} GB/s"
            print(f"{name:28s} {t['calls']:5d} calls {t['ms']:9.2f} ms {100 * t['ms'] / tot:5.1f}% {rate}", file=sys.stderr)
        print(f"{'(sum of entry points)':28s} {tot:22.2f} ms", file=sys.stderr)
    if world == 1 and not a.no_cpu_baseline and a.workload == 'mit_lawin':       # the CPU leg times the headline workload only
        res['cpu_baseline'] = cpu_baseline()
    print(json.dumps(res))
    if comm_on:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
