"""Times the prediction-writer / evaluation kernels at BASELINE configs[3] size (4 x 6 x 2048 x 2048) and the CPU expressions of the
reference beside them (helper/callbacks.py:52-54 through torch, scripts/evaluate.py:116-135 through sklearn) on a bounded sample."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from semseg_amd import ops, postproc


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


B, C, S = 4, 6, 2048
g = torch.Generator(device='cuda').manual_seed(0)
cl = torch.randn((B, S, S, 8), generator=g, device='cuda').to(torch.bfloat16)
nhwc = cl[..., :C].permute(0, 3, 1, 2)                       # what TopFormerModel.forward returns: channels-last, row pitch 8
nchw = nhwc.contiguous()
npix = B * S * S
for name, t, by in (('class_map, channels-last bf16 (pitch 8)', nhwc, npix * (16 + 1)), ('class_map, NCHW bf16', nchw, npix * (12 + 1)),
                    ('class_map, NCHW fp32', nchw.float(), npix * (24 + 1))):
    us = timed(lambda: postproc.class_map(t))
    print(f'{name:44s} {us:8.1f} us  {by / us / 1e6:6.2f} TB/s')
low = torch.randn((B * 512 * 512, 8), generator=g, device='cuda').to(torch.bfloat16)
us_f = timed(lambda: ops.bilinear_class_map(low, B, 512, 512, S, S, C))
us_u = timed(lambda: ops.class_map(ops.bilinear_fwd(low, B, 512, 512, S, S).view(B, S, S, 8)[..., :C].permute(0, 3, 1, 2)))
print(f'{"resize x4 + class map, fused":44s} {us_f:8.1f} us  (writes {npix / 1e6:.1f} MB)   two kernels: {us_u:8.1f} us')
gt = torch.randint(0, 6, (B, S, S), generator=g, device='cuda', dtype=torch.uint8)
pr = postproc.class_map(nhwc)
us = timed(lambda: postproc.confusion_matrix(gt, pr, list(range(6))))
print(f'{"confusion matrix of two uint8 rasters":44s} {us:8.1f} us  {2 * npix / us / 1e6:6.2f} TB/s')
t0 = time.perf_counter(); st = postproc.image_statistics(gt[0], pr[0], list(range(6))); torch.cuda.synchronize()
print(f'image_statistics, one 2048 x 2048 pair (incl. the 392-byte copy to the host): {1e3 * (time.perf_counter() - t0):.2f} ms')
# the reference's CPU expressions on ONE image of the batch
x1 = nchw[:1].float().cpu()
torch.set_num_threads(min(16, os.cpu_count()))
t0 = time.perf_counter(); ref = x1.detach().clone().softmax(dim=1).argmax(dim=1, keepdim=True).numpy(); t_ref = time.perf_counter() - t0
assert np.array_equal(ref[0, 0].astype(np.uint8), pr[0].cpu().numpy())
print(f'reference expression on the CPU, 1 image, {torch.get_num_threads()} threads: {1e3 * t_ref:.1f} ms  (+ {8 * S * S / 1e6:.0f} MB int64 per image over PCIe when run on the device)')
from sklearn.metrics import confusion_matrix, f1_score, jaccard_score
a, b = gt[0].cpu().numpy().reshape(-1), pr[0].cpu().numpy().reshape(-1)
t0 = time.perf_counter()
f1 = f1_score(y_true=a, y_pred=b, labels=list(range(6)), average=None, zero_division=1)
io = jaccard_score(y_true=a, y_pred=b, labels=list(range(6)), average=None, zero_division=1)
cm = confusion_matrix(y_true=a, y_pred=b, labels=list(range(6)), normalize=None)
t_sk = time.perf_counter() - t0
assert np.array_equal(cm, st['cm']) and np.allclose(f1, st['f1'], atol=1e-12) and np.allclose(io, st['iou'], atol=1e-12)
print(f'sklearn f1 + jaccard + confusion_matrix on the CPU, 1 image pair: {1e3 * t_sk:.0f} ms')
# input staging (data_factory/utils.py:6-37) at the headline batch: 16 x 3 x 1024 x 1024
from semseg_amd import datastage
u8 = torch.randint(0, 256, (16, 3, 1024, 1024), generator=g, device='cuda', dtype=torch.uint8)
n = u8.numel()
us = timed(lambda: datastage.image_to_tensor(u8, dtype=torch.bfloat16))
print(f'{"image_to_tensor uint8 -> bf16 channels-last":44s} {us:8.1f} us  {n * 3 / us / 1e6:6.2f} TB/s   (upload {n / 1e6:.0f} MB instead of {4 * n / 1e6:.0f} MB)')
us = timed(lambda: datastage.image_to_tensor(u8, dtype=torch.float32))
print(f'{"image_to_tensor uint8 -> fp32 NCHW":44s} {us:8.1f} us  {n * 5 / us / 1e6:6.2f} TB/s')
lab = torch.randint(0, 6, (16, 1024, 1024), generator=g, device='cuda', dtype=torch.uint8)
us = timed(lambda: datastage.label_to_tensor(lab))
print(f'{"label_to_tensor uint8 -> int64":44s} {us:8.1f} us  {lab.numel() * 9 / us / 1e6:6.2f} TB/s')
x1 = u8[0].cpu().numpy()
t0 = time.perf_counter(); ref = (torch.from_numpy(x1) - torch.zeros(3, 1, 1)) / torch.full((3, 1, 1), 255.0); t_ref = time.perf_counter() - t0
print(f'reference image_to_tensor on the CPU, 1 tile of 3 x 1024 x 1024: {1e3 * t_ref:.2f} ms')
