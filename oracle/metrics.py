"""CPU oracle: confusion-matrix metrics.  TEST INFRASTRUCTURE ONLY.

Restates what /root/reference/helper/metrics.py computes through torchmetrics
(<=0.10 API; helper/metrics.py:44-114).  torchmetrics is absent in the build
image and the reference ships no fixtures for it => "parity unpinned" against
the reference; the formulas follow SURVEY.md §8(c) and are cross-checked in the
tests against sklearn.metrics.confusion_matrix and torch.bincount.
"""
import numpy as np
import torch


def confusion_matrix(preds, target, num_classes):
    """ConfusionMatrix(normalize='none') update (helper/metrics.py:90-95):
    pred = argmax over dim 1 (first max wins), rows = true class, cols = predicted."""
    pred = torch.argmax(preds.float(), dim=1).reshape(-1).long()
    tgt = target.reshape(-1).long()
    cm = torch.bincount(tgt * num_classes + pred, minlength=num_classes ** 2)
    return cm.reshape(num_classes, num_classes).to(torch.int64)


def _div(a, b):
    """a/b with zero-division -> 0 (torchmetrics _safe_divide / zero_division=0)."""
    a = a.astype(np.float32)
    b = b.astype(np.float32)
    out = np.zeros_like(a)
    np.divide(a, b, out=out, where=b != 0)
    return out


def derive(cm):
    """All 8 entries of the reference MetricCollection from one int64 CxC matrix
    (helper/metrics.py:44-114), float32 like torchmetrics."""
    cm = np.asarray(cm, dtype=np.int64)
    tp = np.diag(cm)
    fp = cm.sum(0) - tp
    fn = cm.sum(1) - tp
    absent = (tp + fp + fn) == 0                       # class never seen nor predicted
    prec, rec, f1 = _div(tp, tp + fp), _div(tp, tp + fn), _div(2 * tp, 2 * tp + fp + fn)
    acc = _div(tp, tp + fn)
    for v in (prec, rec, f1, acc):                     # average='none': absent class -> NaN
        v[absent] = np.nan
    union = tp + fp + fn
    iou = _div(tp, union)
    iou[union == 0] = 1.0                              # absent_score=1.0 (helper/metrics.py:66,99)
    return {
        'Precision': prec, 'Recall': rec, 'IoU': iou, 'F1': f1, 'Accuracy': acc,
        'Confusion_Matrix': cm,
        'mIoU': np.float32(iou.mean()),                # macro over ALL classes (:96-103)
        'Global Accuracy': np.float32(np.float32(tp.sum()) / np.float32(cm.sum())) if cm.sum() else np.float32(0),
    }


def loggable(cm, ignore_index=None):
    """helper/metrics.py:152-177 (loggable_dict): Micro_Accuracy = nanmean of the RAW
    confusion-matrix diagonal (sic), vector scores with the ignore class deleted -> nanmean."""
    d = derive(cm)
    out = {'Micro_Accuracy': np.float32(np.nanmean(np.diag(d['Confusion_Matrix']).astype(np.float32)))}
    for k in ('Precision', 'Recall', 'IoU', 'F1', 'Accuracy'):
        v = d[k] if ignore_index is None else np.delete(d[k], ignore_index)
        out[k] = np.float32(np.nanmean(v)) if np.any(~np.isnan(v)) else np.float32(np.nan)
    out['mIoU'], out['Global Accuracy'] = d['mIoU'], d['Global Accuracy']
    return out
