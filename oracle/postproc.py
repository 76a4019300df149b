"""CPU oracle: prediction writer and evaluation-script statistics.  TEST INFRASTRUCTURE ONLY.

class_map restates /root/reference/helper/callbacks.py:52-54 (prediction.softmax(dim=1).argmax(dim=1, keepdim=True)); it is pinned
in tests/test_oracle_golden.py against that very torch expression on CPU.  image_statistics restates what
/root/reference/scripts/evaluate.py:111-156 computes per image pair through sklearn.metrics (f1_score, jaccard_score,
confusion_matrix with `labels`, zero_division=1) plus its own normalised matrix and kappa; sklearn IS importable in this image,
so the restatement is pinned against the library calls themselves.
"""
import numpy as np


def class_map(prediction):
    """prediction: float array [B, C, H, W] -> uint8 [B, H, W].  softmax is increasing, so argmax(softmax(x)) == argmax(x) with
    the first maximum winning (np.argmax, as torch.argmax), except that a NaN anywhere in a pixel makes its whole softmax NaN and
    torch.argmax then returns index 0."""
    p = np.asarray(prediction, dtype=np.float32)
    out = np.argmax(np.where(np.isnan(p), -np.inf, p), axis=1)
    out[np.isnan(p).any(axis=1)] = 0
    return out.astype(np.uint8)


def confusion(gt, pr, label_indices, valid=None):
    """[n + 1, n + 1] int64 counts, rows = truth, index n = "a value that is not among label_indices".  The top-left n x n block is
    sklearn.metrics.confusion_matrix(y_true, y_pred, labels=label_indices) (scripts/evaluate.py:130-135, which drops such pixels);
    the extra row / column keeps what f1_score / jaccard_score still count against the listed classes."""
    gt, pr = np.asarray(gt).reshape(-1), np.asarray(pr).reshape(-1)
    if valid is not None:
        keep = np.asarray(valid).reshape(-1).astype(bool)
        gt, pr = gt[keep], pr[keep]
    n = len(label_indices)
    lut = np.full(256, n, dtype=np.int64)
    for i, v in enumerate(label_indices):
        lut[int(v)] = i
    return np.bincount(lut[gt] * (n + 1) + lut[pr], minlength=(n + 1) ** 2).reshape(n + 1, n + 1).astype(np.int64)


def image_statistics(full):
    """Per-image numbers of scripts/evaluate.py:116-156 from confusion() of that image pair."""
    full = np.asarray(full, dtype=np.int64)
    cm = full[:-1, :-1].copy()
    tp = np.diag(cm).astype(np.float64)
    fp = full[:, :-1].sum(0) - tp
    fn = full[:-1, :].sum(1) - tp
    with np.errstate(divide='ignore', invalid='ignore'):
        f1 = np.where(2 * tp + fp + fn == 0, 1.0, 2 * tp / (2 * tp + fp + fn))          # f1_score(average=None, zero_division=1)
        iou = np.where(tp + fp + fn == 0, 1.0, tp / (tp + fp + fn))                     # jaccard_score(average=None, zero_division=1)
    true_sum = cm.sum(1, keepdims=True)                                                  # :136-141 rows of an absent class stay identity
    norm_cm = np.eye(cm.shape[0])
    np.divide(cm, true_sum, where=(true_sum != 0), out=norm_cm)
    k = cm.copy()                                                                        # :145-155 zeros on the diagonal become ones
    d = k.diagonal().copy()
    np.fill_diagonal(k, d + (d == 0))
    s0, s1 = k.sum(0), k.sum(1)
    expected = np.outer(s0, s1) / s0.sum()
    w = 1 - np.eye(k.shape[0], dtype=np.int64)
    kappa = 1 - (w * k).sum() / (w * expected).sum()
    return {'f1': f1, 'iou': iou, 'cm': cm, 'norm_cm': norm_cm, 'accuracy': norm_cm.diagonal().copy(), 'kappa': float(kappa)}
