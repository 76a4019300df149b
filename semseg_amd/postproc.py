"""The steps right after the forward in predict.py and scripts/evaluate.py, on the device.

* ``class_map`` / ``PredictionWriter``: helper/callbacks.py:20-92 turns every prediction batch into
  ``prediction.softmax(1).argmax(1, keepdim=True).cpu().numpy()`` -- a softmax pass, an int64 class map and a device-to-host copy
  of 8 bytes per pixel.  Here one kernel writes a uint8 class map (1 byte per pixel to the host) straight from the logits.
  ``TopFormerModel.predict_class_map`` goes one step further and never materialises the full-size logits.
* ``image_statistics``: scripts/evaluate.py:111-156 runs three sklearn passes (f1_score, jaccard_score, confusion_matrix) over the
  full-size rasters of every image pair on the CPU.  Here the C x C confusion matrix of the pair is counted on the device
  (exact, int64) and the few numbers derived from it follow the script's own arithmetic in float64.
There is no CPU fallback: tensors must live on a HIP device and the HIP library must load.
"""
import numpy as np
import torch

from . import ops


def class_map(prediction):
    """[B, C, H, W] logits -> uint8 [B, H, W] on the device (argmax over C, first maximum wins; a pixel holding a NaN -> 0)."""
    return ops.class_map(prediction.detach())


class PredictionWriter:
    """helper/callbacks.py:20-92 without the Lightning base class: same constructor, same ``write_on_batch_end`` arguments, same
    ``write_batch(data_batch=[B, 1, H, W], indexes=, overwrite=)`` call on the data set.  ``data_batch`` is uint8 unless
    ``dtype`` asks for the reference's int64."""

    def __init__(self, writable_datasets, overwrite=False, dtype=np.uint8):
        if not isinstance(writable_datasets, (list, tuple)):
            writable_datasets = [writable_datasets]
        assert all(hasattr(w, 'write_batch') for w in writable_datasets), "one or more datasets are not instances of WritableDataset!"
        self.writable_datasets = list(writable_datasets)
        self.overwrite = bool(overwrite)
        self.dtype = np.dtype(dtype)

    def write_on_batch_end(self, trainer, pl_module, prediction, batch_indices, batch, batch_idx, dataloader_idx):
        cls = prediction if prediction.dtype == torch.uint8 else class_map(prediction)       # a model may hand over class maps already
        data_batch = cls.unsqueeze(1).cpu().numpy().astype(self.dtype, copy=False)
        self.writable_datasets[dataloader_idx].write_batch(data_batch=data_batch, indexes=batch_indices, overwrite=self.overwrite)
        return None


_luts = {}


def _lut(label_indices, device):
    key = (tuple(int(v) for v in label_indices), device)
    t = _luts.get(key)
    if t is None:
        if len(set(key[0])) != len(key[0]) or any(v < 0 or v > 255 for v in key[0]):
            raise ValueError('label_indices must be distinct values in 0..255')
        lut = np.full(256, len(key[0]), dtype=np.int32)          # index n = "not among the labels"
        for i, v in enumerate(key[0]):
            lut[v] = i
        t = _luts[key] = torch.from_numpy(lut).to(device)
    return t


def confusion_matrix(gt, pr, label_indices, valid=None):
    """uint8 class maps on the device -> int64 [n + 1, n + 1] tensor, rows = truth, index n = a raster value that is not among
    label_indices.  [:n, :n] is sklearn.metrics.confusion_matrix(gt, pr, labels=label_indices), which drops such pixels; the extra
    row / column keeps what sklearn's f1_score / jaccard_score still count against the listed classes."""
    if valid is not None and valid.dtype == torch.bool:
        valid = valid.view(torch.uint8)
    return ops.confusion_labels(gt, pr, _lut(label_indices, gt.device), len(label_indices) + 1, valid=valid)


def image_statistics(gt, pr, label_indices, valid=None):
    """scripts/evaluate.py:111-156 for one (ground truth, prediction) pair of uint8 rasters on the device; ``valid`` = the script's
    valid_mask (uint8 / bool, None = every pixel).  -> {'f1', 'iou', 'cm', 'norm_cm', 'accuracy', 'kappa'} as numpy / float, from
    the counts of ONE pass over the rasters (the script makes three sklearn passes on the CPU)."""
    full = confusion_matrix(gt, pr, label_indices, valid).cpu().numpy()
    cm = full[:-1, :-1].copy()
    tp = np.diag(cm).astype(np.float64)
    fp, fn = full[:, :-1].sum(0) - tp, full[:-1, :].sum(1) - tp
    with np.errstate(divide='ignore', invalid='ignore'):
        f1 = np.where(2 * tp + fp + fn == 0, 1.0, 2 * tp / (2 * tp + fp + fn))          # f1_score(average=None, zero_division=1)
        iou = np.where(tp + fp + fn == 0, 1.0, tp / (tp + fp + fn))                     # jaccard_score(average=None, zero_division=1)
    true_sum = cm.sum(1, keepdims=True)                                                  # :136-141
    norm_cm = np.eye(cm.shape[0])
    np.divide(cm, true_sum, where=(true_sum != 0), out=norm_cm)
    k = cm.copy()                                                                        # :145-155
    d = k.diagonal().copy()
    np.fill_diagonal(k, d + (d == 0))
    s0, s1 = k.sum(0), k.sum(1)
    off = 1 - np.eye(k.shape[0], dtype=np.int64)
    kappa = 1 - (off * k).sum() / (off * (np.outer(s0, s1) / s0.sum())).sum()
    return {'f1': f1, 'iou': iou, 'cm': cm, 'norm_cm': norm_cm, 'accuracy': norm_cm.diagonal().copy(), 'kappa': float(kappa)}
