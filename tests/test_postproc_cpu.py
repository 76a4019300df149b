"""The oracle for the prediction writer and the evaluation statistics (oracle/postproc.py), pinned against the expressions the
reference itself evaluates: helper/callbacks.py:52-54 through torch on the CPU, scripts/evaluate.py:116-156 through sklearn."""
import numpy as np
import torch

from oracle import postproc as opost


def test_class_map_is_softmax_argmax():
    g = torch.Generator().manual_seed(5)
    for dtype in (torch.float32, torch.bfloat16):
        logits = (torch.randn(3, 6, 17, 23, generator=g) * 3).to(dtype)
        logits[0, :, 3, 4] = 1.25                                  # a full tie: the first class wins
        logits[1, 2, 5, 6] = logits[1, 4, 5, 6] = 9.0              # a two-way tie at the top
        logits[2, 3, 0, 0] = float('nan')                          # softmax turns the pixel into NaN: argmax returns 0
        ref = logits.detach().clone().softmax(dim=1).argmax(dim=1, keepdim=True).cpu().numpy()     # helper/callbacks.py:52-54
        got = opost.class_map(logits.float().numpy())
        if dtype == torch.float32:
            assert np.array_equal(got[:, None], ref)
        else:                              # bf16 softmax rounds distinct logits onto equal probabilities: only those pixels may differ
            p = logits.softmax(dim=1).float().numpy()
            top2 = np.sort(p, axis=1)[:, -2:]
            merged = top2[:, 0] == top2[:, 1]
            assert np.array_equal(got[~merged], ref[:, 0][~merged])


def test_statistics_against_sklearn():
    from sklearn.metrics import confusion_matrix, f1_score, jaccard_score, cohen_kappa_score
    rng = np.random.default_rng(11)
    labels = [0, 1, 2, 3, 4, 5]
    for case in range(4):
        gt = rng.integers(0, 6, size=(64, 48)).astype(np.uint8)
        pr = np.where(rng.random(gt.shape) < 0.7, gt, rng.integers(0, 6, size=gt.shape)).astype(np.uint8)
        if case == 1:                                  # class 4 absent from both rasters, class 5 never predicted
            gt[gt == 4] = 1; pr[pr == 4] = 1; pr[pr == 5] = 0
        if case == 2:                                  # values outside `labels` are dropped
            gt[:4] = 9; pr[-3:] = 200
        valid = rng.random(gt.shape) < 0.9 if case == 3 else None
        a, b = (gt[valid], pr[valid]) if valid is not None else (gt.reshape(-1), pr.reshape(-1))
        full = opost.confusion(gt, pr, labels, valid)
        cm = full[:6, :6]
        assert np.array_equal(cm, confusion_matrix(y_true=a, y_pred=b, labels=labels, normalize=None))
        assert full.sum() == a.size and (case == 2) == bool(full[6].sum() + full[:, 6].sum())
        st = opost.image_statistics(full)
        # with values outside `labels` (case 2) sklearn's matrix drops the pixel, its f1 / jaccard still count it against the listed class
        np.testing.assert_allclose(st['f1'], f1_score(y_true=a, y_pred=b, labels=labels, average=None, zero_division=1), rtol=0, atol=1e-12)
        np.testing.assert_allclose(st['iou'], jaccard_score(y_true=a, y_pred=b, labels=labels, average=None, zero_division=1), rtol=0, atol=1e-12)
        if case in (0, 3):                             # no zero on the diagonal: the script's kappa is Cohen's kappa
            assert abs(st['kappa'] - cohen_kappa_score(a, b, labels=labels)) < 1e-12
        ts = cm.sum(1)
        for i in range(6):
            row = cm[i] / ts[i] if ts[i] else np.eye(6)[i]
            np.testing.assert_allclose(st['norm_cm'][i], row, rtol=0, atol=0)


def test_scores_from_confmat_exact_beyond_2_24():
    """derived scores from counts that float32 cannot hold (one 16 x 1024^2 step already has 2^24 pixels): integer sums, one
    rounding at the end (the counts used to be converted to float32 before the sums)."""
    from fractions import Fraction
    import torch
    from semseg_amd.metrics import scores_from_confmat
    cm = torch.tensor([[2 ** 33 + 1, 3, 0], [5, 2 ** 25 + 3, 0], [0, 0, 0]], dtype=torch.int64)
    out = scores_from_confmat(cm)
    c = cm.tolist()
    tp = [c[i][i] for i in range(3)]
    fp = [sum(c[r][i] for r in range(3)) - tp[i] for i in range(3)]
    fn = [sum(c[i]) - tp[i] for i in range(3)]
    f32 = lambda fr: float(torch.tensor(float(fr), dtype=torch.float64).float())
    for i in range(2):
        assert float(out['Precision'][i]) == f32(Fraction(tp[i], tp[i] + fp[i]))
        assert float(out['Recall'][i]) == f32(Fraction(tp[i], tp[i] + fn[i]))
        assert float(out['IoU'][i]) == f32(Fraction(tp[i], tp[i] + fp[i] + fn[i]))
        assert float(out['F1'][i]) == f32(Fraction(2 * tp[i], 2 * tp[i] + fp[i] + fn[i]))
    assert torch.isnan(out['Precision'][2]) and float(out['IoU'][2]) == 1.0          # absent class: NaN / absent_score 1.0
    assert float(out['Global Accuracy']) == f32(Fraction(sum(tp), sum(map(sum, c))))
    assert out['Confusion_Matrix'].dtype == torch.int64
