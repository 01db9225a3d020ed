"""Epoch-level binary classification report of `MultiModN.test()` (reference:
multimodn/multimodn.py:18-49 `get_performance_metrics`), without the torchmetrics dependency.

The reference delegates to torchmetrics (F1Score / ROC / PrecisionRecallCurve / Accuracy / AUROC /
ConfusionMatrix, task="binary"); that package is unpinned in the reference's requirements and not
part of this image, so its published exact-mode (thresholds=None) algorithms are restated here with
torch ops: one sort of the scores by descending value, cumulative sums, one curve point per distinct
score.  This runs once per `test()` call on N scores per decoder: epoch-level post-processing, not
part of the per-batch hot path.
"""
from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor

performance_metrics = ['f1', 'auc', 'accuracy', 'sensitivity', 'specificity', 'fpr', 'tpr', 'precision', 'recall',
                       'tn', 'fp', 'fn', 'tp', 'thr_roc', 'thr_pr']          # multimodn.py:18-19


def _clf_curve(y_prob: Tensor, y_true: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """(fps, tps, thresholds) at every distinct score, scores descending."""
    order = torch.argsort(y_prob, descending=True, stable=True)
    ps, ts = y_prob[order], y_true[order].to(torch.float64)
    n = ps.numel()
    if n == 0:
        z = torch.zeros(0, dtype=torch.float64, device=y_prob.device)
        return z, z, ps
    distinct = torch.nonzero(ps[1:] != ps[:-1]).flatten()
    idx = torch.cat([distinct, torch.tensor([n - 1], device=ps.device)])
    tps = torch.cumsum(ts, 0)[idx]
    fps = (1 + idx).to(torch.float64) - tps
    return fps, tps, ps[idx]


def compute_metrics(tp, tn, fp, fn, cm, enc_idx, dec_idx):
    """Adds one binary confusion matrix cm[true][pred] to the (E+1) x D counter grids at [enc_idx][dec_idx]; cm = None (more
    than two classes) marks the cell NaN (multimodn.py:51-63).  train_epoch / test do not call it - the decoder-grid kernel
    counts - it is here for code that imports it from the reference's module."""
    if cm is not None:
        tp[enc_idx][dec_idx] += cm[1][1]
        tn[enc_idx][dec_idx] += cm[0][0]
        fp[enc_idx][dec_idx] += cm[0][1]
        fn[enc_idx][dec_idx] += cm[1][0]
    else:
        for grid in (tp, tn, fp, fn):
            grid[enc_idx][dec_idx] = float('nan')


def get_performance_metrics(y_true: Tensor, y_pred: Tensor, y_prob: Tensor):
    """Same 15-tuple, in the same order, as the reference's get_performance_metrics (:47-49):
    (f1, auc, accuracy, sensitivity, specificity, fpr, tpr, precision, recall, tn, fp, fn, tp,
    thr_roc, thr_pr).  Scalars are 0-dim tensors (python 0 for an undefined sensitivity /
    specificity, as in the reference :38-45)."""
    y_true = y_true.to(torch.int64).flatten()
    y_pred = y_pred.to(torch.int64).flatten()
    y_prob = y_prob.to(torch.float32).flatten()
    # ConfusionMatrix(task="binary")(y_pred, y_true) -> cm[true][pred]  (:29-33)
    tp = ((y_pred == 1) & (y_true == 1)).sum()
    tn = ((y_pred == 0) & (y_true == 0)).sum()
    fp = ((y_pred == 1) & (y_true == 0)).sum()
    fn = ((y_pred == 0) & (y_true == 1)).sum()
    sensitivity = tp / (tp + fn) if int(tp + fn) != 0 else 0
    specificity = tn / (tn + fp) if int(tn + fp) != 0 else 0
    # F1Score(task="binary") on probabilities: thresholded at 0.5
    hard = (y_prob > 0.5).to(torch.int64)
    tp5 = ((hard == 1) & (y_true == 1)).sum().to(torch.float32)
    fp5 = ((hard == 1) & (y_true == 0)).sum().to(torch.float32)
    fn5 = ((hard == 0) & (y_true == 1)).sum().to(torch.float32)
    den = 2 * tp5 + fp5 + fn5
    f1 = torch.where(den == 0, torch.zeros_like(den), 2 * tp5 / torch.clamp(den, min=1))
    accuracy = (y_pred == y_true).to(torch.float32).mean() if y_true.numel() else torch.tensor(0.0)
    fps, tps, thr = _clf_curve(y_prob, y_true)
    zero = torch.zeros(1, dtype=fps.dtype, device=fps.device)
    fps0, tps0 = torch.cat([zero, fps]), torch.cat([zero, tps])
    fpr = fps0 / fps0[-1] if fps0.numel() > 1 and float(fps0[-1]) > 0 else torch.zeros_like(fps0)
    tpr = tps0 / tps0[-1] if tps0.numel() > 1 and float(tps0[-1]) > 0 else torch.zeros_like(tps0)
    thr_roc = torch.cat([torch.ones(1, dtype=thr.dtype, device=thr.device), thr])
    auc = torch.trapz(tpr, fpr).to(torch.float32)
    precision = tps / (tps + fps)
    recall = tps / tps[-1] if tps.numel() and float(tps[-1]) > 0 else torch.zeros_like(tps)
    one = torch.ones(1, dtype=precision.dtype, device=precision.device)
    precision = torch.cat([precision.flip(0), one])
    recall = torch.cat([recall.flip(0), zero])
    thr_pr = thr.flip(0).clone()
    return (f1, auc, accuracy, sensitivity, specificity, fpr.to(torch.float32), tpr.to(torch.float32),
            precision.to(torch.float32), recall.to(torch.float32), tn, fp, fn, tp, thr_roc, thr_pr)
