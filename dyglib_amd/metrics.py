"""Evaluation metrics on the device: the reference's utils/metrics.py (which copies the scores to the host and calls
scikit-learn) behind the same two functions, plus a batched form that leaves the results on the GPU so an evaluation
loop never synchronises per batch.  The math is dygnn_link_metrics (dyglib_amd/csrc/metrics.hip);
there is no host fallback."""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from . import _capi


def link_prediction_metrics_device(predicts: torch.Tensor, labels: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """predicts, labels: [n_groups, group_size] (or [group_size]) float tensors on the GPU; one group = one call of the
    reference's get_link_prediction_metrics (utils/metrics.py:5-19) plus its BCELoss (evaluate_models_utils.py:145).
    -> (average_precision, roc_auc, bce_loss, status), each [n_groups] on the device (float64 x3, int32); nothing is
    synchronised.  status 1 = only one class in the group (roc_auc is NaN there; scikit-learn raises ValueError)."""
    if not predicts.is_cuda or not labels.is_cuda:
        raise _capi.DygnnError("metrics run on the GPU only (tensors are on %s / %s)" % (predicts.device, labels.device))
    if predicts.shape != labels.shape or predicts.dim() not in (1, 2):
        raise AssertionError("predicts and labels must have the same shape [n_groups, group_size] or [group_size]")
    p = predicts.detach().reshape(-1, predicts.shape[-1]).contiguous().float()
    y = labels.detach().reshape(-1, labels.shape[-1]).contiguous().float()
    G, n = p.shape
    if n == 0:
        raise ValueError("Found array with 0 sample(s) while a minimum of 1 is required.")      # scikit-learn's check_array
    lib = _capi.load()
    dev = p.device
    ap = torch.empty(G, dtype=torch.float64, device=dev)
    auc = torch.empty(G, dtype=torch.float64, device=dev)
    loss = torch.empty(G, dtype=torch.float64, device=dev)
    status = torch.empty(G, dtype=torch.int32, device=dev)
    ws_bytes = lib.dygnn_link_metrics_workspace_bytes(n, G)
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=dev)
    _capi.check(lib.dygnn_link_metrics(p.data_ptr(), y.data_ptr(), n, G, ap.data_ptr(), auc.data_ptr(), loss.data_ptr(),
                                       status.data_ptr(), ws.data_ptr(), ws_bytes, _capi.current_stream_ptr()))
    return ap, auc, loss, status


def get_link_prediction_metrics(predicts: torch.Tensor, labels: torch.Tensor) -> Dict[str, float]:
    """utils/metrics.py:5-19: {'average_precision', 'roc_auc'} as Python floats for predicts / labels of shape (num_samples,)."""
    ap, auc, _, status = link_prediction_metrics_device(predicts.reshape(1, -1), labels.reshape(1, -1))
    if int(status.item()) != 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    return {"average_precision": float(ap.item()), "roc_auc": float(auc.item())}


def get_node_classification_metrics(predicts: torch.Tensor, labels: torch.Tensor) -> Dict[str, float]:
    """utils/metrics.py:22-34: {'roc_auc'} for predicts / labels of shape (num_samples,)."""
    _, auc, _, status = link_prediction_metrics_device(predicts.reshape(1, -1), labels.reshape(1, -1))
    if int(status.item()) != 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    return {"roc_auc": float(auc.item())}
