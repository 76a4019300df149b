"""Losses with the reference interfaces (loss/seg_loss.py, helper/weights.py) on the fused HIP kernels."""
from typing import Sequence, Union
import torch
from torch import Tensor
from torch.nn import Module

from . import ops

__all__ = ['DiceLoss', 'OhemCrossEntropyLoss', 'dice_loss', 'class_weights']


class _OhemFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, preds, targets, ignore_index, min_kept, thr_state):
        loss, prob, ctl = ops.ohem_ce_fwd(preds.detach(), targets, ignore_index, min_kept, thr_state)
        ctx.save_for_backward(preds, targets)
        ctx.prob, ctx.ctl, ctx.ignore = prob, ctl, ignore_index
        return loss

    @staticmethod
    def backward(ctx, gout):
        preds, targets = ctx.saved_tensors
        d = ops.ohem_ce_bwd(preds.detach(), targets, ctx.prob, ctx.ctl, gout, ctx.ignore)
        return d, None, None, None, None


class OhemCrossEntropyLoss(Module):
    """loss/seg_loss.py:103-167.  The stateful threshold (:143-144) lives in a device scalar so no step synchronises the
    host; ``.threshold`` reads it back on demand."""

    def __init__(self, ignore_index: int, threshold: float = 0.7, min_kept: float = 0.1) -> None:
        super().__init__()
        self.ignore_index = ignore_index
        assert 0 <= min_kept <= 1.0, f"In valid 'min_kept' ratio: {min_kept}.\nValid range: [0.0, 1.0]"
        self.min_kept = min_kept
        self._thr0 = float(threshold)
        self._thr = None

    @property
    def threshold(self):
        return self._thr0 if self._thr is None else float(self._thr)

    @threshold.setter
    def threshold(self, value):
        self._thr0, self._thr = float(value), None

    def forward(self, preds: Tensor, targets: Tensor) -> Tensor:
        if self._thr is None or self._thr.device != preds.device:
            self._thr = torch.full((1,), self._thr0, dtype=torch.float32, device=preds.device)
        return _OhemFn.apply(preds, targets.contiguous(), self.ignore_index, self.min_kept, self._thr)


def class_weights(x: torch.Tensor, class_ids: torch.Tensor):
    """helper/weights.py:4-27: inverse-frequency weights of the classes ``class_ids`` in the label tensor ``x`` (any integer
    values in 0..4095; an id that never occurs gets weight 0), from one label-histogram pass."""
    ids = class_ids.reshape(-1).to(device=x.device, dtype=torch.long)
    if ids.numel() == 0:
        raise ValueError('class_weights: class_ids is empty')
    lo, hi = int(ids.min()), int(ids.max())
    if lo < 0 or hi >= 4096:
        raise ValueError(f'class_weights: class ids must lie in 0..4095 (got {lo}..{hi})')
    hist = ops.label_hist(x.reshape(-1).to(torch.int64).contiguous(), hi + 1)
    return _weights_from_freq(hist[ids].to(torch.float32))


def _weights_from_freq(freq):
    """frequencies (float32; exact below 2^24 per class, as the reference's integer / integer true division rounds) -> weights"""
    w = freq.sum() / (freq.numel() * freq)
    w = torch.where(freq == 0, torch.zeros_like(w), w)
    return w / w.sum()


class _DiceFn(torch.autograd.Function):
    """mode 'tversky' = DiceLoss.forward (:82-99); mode 'dice' = dice_loss (:15-55).  The [B,C] coefficient algebra between
    the two kernels runs on B*C-element tensors."""

    @staticmethod
    def forward(ctx, preds, labels, mode, a, b, weighted):
        sums = ops.dice_sums(preds.detach(), labels)            # [B, C, 3]: tp, sum p, count
        tp, sp, cnt = sums[..., 0], sums[..., 1], sums[..., 2]
        B, C = tp.shape
        if mode == 'tversky':
            delta = a
            den = delta * cnt + (1 - delta) * sp + 1e-6          # tp + d*fn + (1-d)*fp + eps with fn = cnt-tp, fp = sp-tp
            score = (tp + 1e-6) / den
            w = _weights_from_freq(cnt.sum(0)) if weighted else torch.ones(C, device=tp.device)
            loss = (w * (1 - score.mean(0))).mean()
            k = -w / (B * C)
            coef_hit = (k / den)
            coef_all = -(k * (tp + 1e-6) * (1 - delta) / den ** 2)
        else:
            smooth, eps = a, b
            m, n = (smooth, smooth) if smooth else (0.0, eps)
            inter, card = tp.sum(0), (sp + cnt).sum(0)
            loss = (1.0 - (2.0 * inter + m) / (card + n)).mean()
            coef_hit = (-2.0 / (card + n) / C).expand(B, C)
            coef_all = ((2.0 * inter + m) / (card + n) ** 2 / C).expand(B, C)
        ctx.save_for_backward(preds, labels)
        ctx.coefs = (coef_hit.contiguous().float(), coef_all.contiguous().float())
        return loss

    @staticmethod
    def backward(ctx, gout):
        preds, labels = ctx.saved_tensors
        return ops.dice_bwd(preds, labels, ctx.coefs[0], ctx.coefs[1], gout), None, None, None, None, None


def dice_loss(preds: torch.Tensor, targets: torch.Tensor, smooth: float = 0.0, eps: float = 1e-7):
    """loss/seg_loss.py:15-55 (multi-class form)."""
    if preds.size(1) == 1:
        raise NotImplementedError('dice_loss: the single-channel (sigmoid) form is not on the MI355X path')
    return _DiceFn.apply(preds, targets.contiguous(), 'dice', float(smooth), float(eps), False)


class DiceLoss(Module):
    def __init__(self, delta: float = 0.5, weighted: bool = False, class_ids=Union[int, Sequence[int]]):
        super().__init__()
        self.delta = delta
        if isinstance(class_ids, int):
            self.class_ids = torch.arange(class_ids, dtype=torch.long)
        else:
            assert isinstance(class_ids, Sequence) and all([isinstance(i, int) for i in class_ids]), \
                "'class_ids' are not a integer or a sequence of integer!"
            self.class_ids = torch.tensor(class_ids, dtype=torch.long)
        self.weighted = weighted
        self.num_classes = self.class_ids.numel()

    def forward(self, preds: Tensor, labels: Tensor) -> Tensor:
        if preds.shape[1] != self.num_classes or not torch.equal(self.class_ids, torch.arange(self.num_classes)):
            raise NotImplementedError('DiceLoss: class_ids must be arange(num_classes) == channels of preds')
        return _DiceFn.apply(preds, labels.contiguous(), 'tversky', float(self.delta), 0.0, bool(self.weighted))


class _Softmax2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = ops.softmax2d_fwd(x.detach())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.softmax2d_bwd(y, dy if dy.dtype == y.dtype else dy.to(y.dtype))


class Softmax2d(Module):
    """torch.nn.Softmax2d (the last layer of modules/layers.py:25-43) on the HIP path, for logits that have no fused producer."""

    def forward(self, x: Tensor) -> Tensor:
        return _Softmax2dFn.apply(x)


class OhemCEDiceLoss(Module):
    """OHEM cross-entropy on the logits plus Tversky / Dice on their softmax: the "CE + Dice" objective BASELINE configs[4] names
    for MiT-B5 + Lawin.  The reference holds the two terms as separate classes (loss/seg_loss.py:59-167) and its networks that feed
    DiceLoss end in Softmax2d; this is their sum, ``ce_weight * OhemCrossEntropyLoss + dice_weight * DiceLoss(Softmax2d(preds))``."""

    def __init__(self, num_classes: int, ignore_index: int, threshold: float = 0.7, min_kept: float = 0.1, delta: float = 0.5,
                 weighted: bool = False, ce_weight: float = 1.0, dice_weight: float = 1.0):
        super().__init__()
        self.ce = OhemCrossEntropyLoss(ignore_index, threshold, min_kept)
        self.dice = DiceLoss(delta, weighted, num_classes)
        self.softmax = Softmax2d()
        self.ce_weight, self.dice_weight = ce_weight, dice_weight

    def forward(self, preds: Tensor, targets: Tensor) -> Tensor:
        return self.ce_weight * self.ce(preds, targets) + self.dice_weight * self.dice(self.softmax(preds), targets)
