"""CPU restatement of the evaluation metrics of the reference's utils/metrics.py:5-34.  TEST INFRASTRUCTURE ONLY (see
oracle/dygformer_oracle.py header): imported by tests/ and by bench.py's cpu_baseline leg, never by the product.

The reference delegates to scikit-learn (third-party, not vendored; requirements.txt lists `scikit_learn` unpinned, 1.7.2 in
this image): `average_precision_score` and `roc_auc_score`.  This file restates their published algorithm for the binary
case (sklearn/metrics/_ranking.py: `_binary_clf_curve`, `precision_recall_curve`, `roc_curve`, `auc`): stable sort by
descending score, one threshold per distinct score, cumulative true/false positives, then the step integral (AP) and the
trapezoid rule (AUC).  Pinned by tests/golden/metrics.npz, which oracle/make_golden.py produced by calling the reference's
own get_link_prediction_metrics / get_node_classification_metrics.  The HIP kernel uses a different (sort-free, counting)
formulation, so agreement between the two is a real check."""
from __future__ import annotations

import numpy as np


def _binary_clf_curve(y_true: np.ndarray, y_score: np.ndarray):
    order = np.argsort(y_score, kind="mergesort")[::-1]
    y_score, y_true = y_score[order], y_true[order]
    distinct = np.where(np.diff(y_score))[0]
    idx = np.r_[distinct, y_true.size - 1]
    tps = np.cumsum(y_true.astype(np.float64))[idx]
    fps = 1 + idx - tps
    return fps, tps


def average_precision(y_true: np.ndarray, y_score: np.ndarray) -> float:
    fps, tps = _binary_clf_curve(y_true != 0, y_score)
    precision = tps / (tps + fps)
    recall = tps / tps[-1]
    prev = np.r_[0.0, recall[:-1]]
    return float(np.sum((recall - prev) * precision))


def roc_auc(y_true: np.ndarray, y_score: np.ndarray) -> float:
    pos = y_true != 0
    if pos.all() or not pos.any():
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    fps, tps = _binary_clf_curve(pos, y_score)
    fpr = np.r_[0.0, fps / fps[-1]]
    tpr = np.r_[0.0, tps / tps[-1]]
    return float(np.sum(np.diff(fpr) * (tpr[1:] + tpr[:-1]) * 0.5))


def bce_loss(y_true: np.ndarray, y_score: np.ndarray) -> float:
    """torch.nn.BCELoss (mean): float32 elementwise, logs clamped at -100."""
    p = y_score.astype(np.float32)
    y = y_true.astype(np.float32)
    with np.errstate(divide="ignore"):
        lp = np.maximum(np.log(p), np.float32(-100.0))
        lq = np.maximum(np.log(np.float32(1.0) - p), np.float32(-100.0))
    return float(np.mean((-(y * lp + (1 - y) * lq)).astype(np.float64)))


def get_link_prediction_metrics(predicts: np.ndarray, labels: np.ndarray) -> dict:
    return {"average_precision": average_precision(labels, predicts), "roc_auc": roc_auc(labels, predicts)}
