"""CPU oracle: OHEM cross-entropy, Dice/Tversky losses, class weights.
TEST INFRASTRUCTURE ONLY.  Follows /root/reference/loss/seg_loss.py and
/root/reference/helper/weights.py.
"""
import torch
import torch.nn.functional as F


class OhemCE:
    """loss/seg_loss.py:103-167, including the stateful, never-decreasing
    threshold (:143-144) and the quirk that the argsort index over the
    *valid-only* probabilities is applied to the *all-pixel* vector (:141-143)."""

    def __init__(self, ignore_index, threshold=0.7, min_kept=0.1):
        self.ignore_index, self.threshold, self.min_kept = ignore_index, threshold, min_kept

    def __call__(self, preds, targets):
        n, c, h, w = preds.shape
        label = targets.detach().clone().reshape(-1)
        valid = label != self.ignore_index
        num_valid = int(valid.sum())
        with torch.no_grad():
            prob = preds.detach().softmax(dim=1).permute(1, 0, 2, 3).reshape(c, -1)
            prob = prob.gather(0, label.view(1, -1)).view(-1)          # :132-135
        valid_prob = prob[valid]
        min_kept = int(self.min_kept * valid_prob.numel())            # :138
        if min_kept < num_valid and num_valid > 0 and min_kept > 0:   # :139-140
            index = valid_prob.argsort(dim=-1, stable=True)
            ti = index[min(index.numel(), min_kept) - 1]
            if prob[ti] > self.threshold:                             # :143 (sic: prob, not valid_prob)
                self.threshold = prob[ti]
            valid = valid & (prob < self.threshold)
        label[~valid] = self.ignore_index
        loss = F.cross_entropy(preds, label.view(n, h, w), ignore_index=self.ignore_index,
                               reduction='none')
        return loss[valid.view(n, h, w)].mean()


def class_weights(x, class_ids):
    """helper/weights.py:4-27."""
    x = x.reshape(-1).long()
    ids = class_ids.reshape(-1).long()
    freq = torch.stack([(x == i).sum() for i in ids])
    wts = freq.sum() / (ids.numel() * freq)
    wts[freq == 0] = 0
    return wts / wts.sum()


def tversky_dice(preds, labels, num_classes, delta=0.5, weighted=False):
    """loss/seg_loss.py:82-99 (DiceLoss.forward); class_ids = arange(num_classes)."""
    wts = class_weights(labels, torch.arange(num_classes)) if weighted else 1
    oh = F.one_hot(labels, num_classes).permute(0, 3, 1, 2)
    tp = (oh * preds).sum(dim=(2, 3))
    fn = (oh * (1 - preds)).sum(dim=(2, 3))
    fp = ((1 - oh) * preds).sum(dim=(2, 3))
    score = (tp + 1e-6) / (tp + delta * fn + (1 - delta) * fp + 1e-6)
    return (wts * (1 - score.mean(dim=0))).mean()


def dice_functional(preds, targets, smooth=0.0, eps=1e-7):
    """loss/seg_loss.py:15-55 (multi-class branch; C==1 builds [p, 1-p])."""
    nc = preds.size(1)
    oh = F.one_hot(targets, nc).type_as(preds).permute(0, 3, 1, 2)
    probas = torch.cat([preds, 1 - preds], dim=1) if nc == 1 else preds
    inter = (probas * oh).sum(dim=(0, 2, 3))
    card = (probas + oh).sum(dim=(0, 2, 3))
    m, n = (smooth, smooth) if smooth else (0, eps)
    return (1.0 - (2.0 * inter + m) / (card + n)).mean()
