"""SegmentationMetrics with the reference interface (helper/metrics.py:16-245) on ONE exact int64 confusion matrix.

The reference builds a torchmetrics MetricCollection of 8 metrics (Precision, Recall, IoU, F1, Accuracy, Confusion_Matrix,
mIoU, Global Accuracy; :44-114) that each re-derive their statistics from the same predictions on every step.  Here
``update`` is one HIP kernel (arg-max over classes + LDS-privatised integer histogram, ``ops.confusion_update``) and every
score is derived from the [C, C] matrix at ``compute`` time.  Across ranks the matrix is summed with one int64 all-reduce
(RCCL), so multi-GPU results are bit-identical to a single-GPU pass over the union of tiles.
"""
from typing import Sequence, Union
import torch
from torch import nn

from . import ops


def delete_indices(tensor: torch.Tensor, indices):
    """helper/utils.py:127-137."""
    if indices is None:
        return tensor
    mask = torch.ones_like(tensor, dtype=torch.bool)
    mask[indices] = False
    return tensor[mask]


def scores_from_confmat(cm: torch.Tensor):
    """All derived scores (float32) from an int64 [C, C] matrix (rows = target, cols = prediction).  Sums and differences stay
    in int64 (exact at any pixel count); only the final ratios are taken in floating point: the statistics are converted once, as
    torchmetrics' ``_reduce_stat_scores`` does with its integer tp / fp / fn, and divided in float64 so that a count above 2^24 is
    not rounded before the division; the result is rounded to float32 once."""
    cm = cm.to(torch.int64)
    tp = cm.diagonal()
    fp = cm.sum(0) - tp
    fn = cm.sum(1) - tp
    union = tp + fp + fn
    absent = union == 0
    f64 = torch.float64
    nan = torch.full(tp.shape, float('nan'), dtype=f64, device=cm.device)

    def ratio(a, b):
        a, b = a.to(f64), b.to(f64)
        return torch.where(b != 0, a / torch.where(b != 0, b, torch.ones_like(b)), torch.zeros_like(a))

    prec = torch.where(absent, nan, ratio(tp, tp + fp)).float()
    rec = torch.where(absent, nan, ratio(tp, tp + fn)).float()
    f1 = torch.where(absent, nan, ratio(2 * tp, 2 * tp + fp + fn)).float()
    iou64 = torch.where(absent, torch.ones_like(nan), ratio(tp, union))                       # absent_score = 1.0
    return {'Precision': prec, 'Recall': rec, 'IoU': iou64.float(), 'F1': f1, 'Accuracy': rec.clone(), 'Confusion_Matrix': cm,
            'mIoU': iou64.mean().float(), 'Global Accuracy': ratio(tp.sum(), cm.sum()).float()}


class SegmentationMetrics(nn.Module):
    supported_norms = {'none', 'true', 'pred', 'all'}

    def __init__(self, num_classes: int, multiclass: bool = True, ignore_index: int = None, prefix: str = None,
                 postfix: str = None, normalization: str = 'true', delimiter: str = '_'):
        super().__init__()
        assert normalization in self.supported_norms, f"Invalid 'normalization': {normalization}!\nSupported: {self.supported_norms}"
        self.normalize = normalization
        self.num_classes = num_classes
        if ignore_index:
            assert 0 <= ignore_index < self.num_classes, "'ignore_index' is out of range!"
        self.ignore_index = ignore_index
        self.pre, self.post = prefix, postfix
        self.prefix = f"{prefix}{delimiter}" if prefix else None
        self.postfix = f"{delimiter}{postfix}" if postfix else None
        self.vector_keys = ("Precision", "Recall", "IoU", "F1", "Accuracy")
        self.scalar_keys = ("mIoU", "Global Accuracy")
        self.cm_key = "Confusion_Matrix"
        self.register_buffer('confmat', torch.zeros(num_classes, num_classes, dtype=torch.int64), persistent=False)

    # ---- state
    def update(self, preds: torch.Tensor, target: torch.Tensor):
        if preds.dim() == 4:
            ops.confusion_update(preds.detach(), target.contiguous(), self.confmat)
        elif preds.shape == target.shape and not preds.is_floating_point():    # already class indices [B, H, W]
            ops.confusion_indices(preds.to(torch.int64).contiguous(), target.to(torch.int64).contiguous(), self.confmat)
        else:
            raise ValueError('SegmentationMetrics.update expects [B, C, H, W] scores or integer class indices shaped like target')

    def reset(self):
        self.confmat.zero_()

    def synced_confmat(self):
        cm = self.confmat
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            from .engine import all_reduce_sum
            cm = cm.clone()
            all_reduce_sum(cm)
        return cm

    def compute(self):
        out = scores_from_confmat(self.synced_confmat())
        return {self.wrap_keys(k): v for k, v in out.items()}

    def forward(self, preds, target):
        self.update(preds, target)
        return self.compute()

    # ---- reference reporting helpers (helper/metrics.py:134-245)
    def wrap_keys(self, keys=Union[str, Sequence[str]]):
        pre = f"{self.prefix}" if self.prefix else ''
        suf = f"{self.postfix}" if self.postfix else ''
        if isinstance(keys, str):
            return f"{pre}{keys}{suf}"
        assert isinstance(keys, Sequence), "'keys' argument is not a string or Sequence"
        return [f"{pre}{k}{suf}" for k in keys]

    def loggable_dict(self):
        class_scores = self.compute()
        score_dict = dict()
        score_dict[self.wrap_keys('Micro_Accuracy')] = torch.nanmean(1.0 * torch.diagonal(class_scores[self.wrap_keys(self.cm_key)]))
        for k in self.wrap_keys(self.vector_keys):
            score_dict[k] = (1.0 * delete_indices(class_scores[k], self.ignore_index)).nanmean()
        for k in self.wrap_keys(self.scalar_keys):
            score_dict[k] = class_scores[k]
        return score_dict

    def tabular_report(self):
        import pandas as pd
        score_dict = self.compute()
        data = torch.stack([delete_indices(score_dict[self.wrap_keys(k)], self.ignore_index) for k in self.vector_keys], dim=-1).tolist()
        indexes = [f"C_{i}" for i in range(self.num_classes)]
        true_indexes = [a for i, a in enumerate(indexes) if i not in {self.ignore_index}]
        class_df = pd.DataFrame(data=data, index=true_indexes, columns=self.vector_keys)
        class_df.index.name = r"Class \ Metric"
        scalar_df = pd.DataFrame(data=[[score_dict[k].item() for k in self.wrap_keys(self.scalar_keys)]], index=['Score'],
                                 columns=self.scalar_keys)
        confmat = score_dict[self.wrap_keys(self.cm_key)]
        if self.normalize == "true":
            confmat = confmat / confmat.sum(axis=1, keepdim=True)
        elif self.normalize == "pred":
            confmat = confmat / confmat.sum(axis=0, keepdim=True)
        elif self.normalize == "all":
            confmat = confmat / confmat.sum()
        confmat_df = pd.DataFrame(data=confmat.tolist(), index=indexes, columns=indexes)
        scalar_df.index.name = r"True \ Predicted"
        return {'prefix': self.pre, 'postfix': self.post, 'class_report': class_df, 'quality_report': scalar_df,
                'confusion_matrix': confmat_df}
