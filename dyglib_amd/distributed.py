"""Multi-GPU evaluation of the link-prediction forward path: one process per GPU, launched with
torch.distributed.run, backend "nccl" (= RCCL over xGMI on ROCm) — or "gloo" on CPU for the tests.

The path shards by WHOLE batches: evaluation batches are independent (the graph,
feature tables and weights are read-only and `recent` sampling is stateless), so batch i goes to rank
i mod world, the CSR + tables + weights are replicated per GPU, and there is NO data-path collective.
A batch is never split across GPUs: an example's output depends on its batch through the padded
lengths S_src/S_dst (models/DyGFormer.py:219-226), so splitting would change the numerics.  The only
exchange is the reduction of the per-batch metric sums — 3 float64 scalars per reduction, latency-bound,
nowhere near the per-link xGMI bandwidth.

The reference computes per-batch AP / AUC with scikit-learn on the host (utils/metrics.py:5-19, one
device->host sync per batch).  `binary_auc` / `average_precision` below are the same estimators on the
device (tie handling identical to sklearn's), so a step needs no host round trip.
"""
from __future__ import annotations

from typing import Callable, Iterable, Optional, Tuple

import torch
import torch.distributed as dist


# ---------------------------------------------------------------------------------------------------
# sharding
# ---------------------------------------------------------------------------------------------------
def shard_batch_indices(num_batches: int, rank: int, world_size: int) -> range:
    """Batches of this rank: i = rank, rank + world, ...  (whole batches, round-robin)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return range(rank, num_batches, world_size)


def owner_of_batch(batch_index: int, world_size: int) -> int:
    return batch_index % world_size


# ---------------------------------------------------------------------------------------------------
# metrics on the device (utils/metrics.py:5-19 with labels = [1]*len(pos) + [0]*len(neg))
# ---------------------------------------------------------------------------------------------------
def binary_auc(pos: torch.Tensor, neg: torch.Tensor) -> torch.Tensor:
    """roc_auc_score for scores cat(pos,neg): P(pos > neg) + 0.5 P(pos == neg) (the Mann-Whitney statistic,
    which is what sklearn's trapezoidal ROC integral equals, ties included)."""
    p, n = pos.double().reshape(-1, 1), neg.double().reshape(1, -1)
    return ((p > n).double().mean() + 0.5 * (p == n).double().mean())


def average_precision(pos: torch.Tensor, neg: torch.Tensor) -> torch.Tensor:
    """average_precision_score: sum_k (R_k - R_{k-1}) P_k over the DISTINCT score thresholds in
    descending order (sklearn groups tied scores into one threshold)."""
    scores = torch.cat([pos.reshape(-1), neg.reshape(-1)]).double()
    labels = torch.cat([torch.ones(pos.numel(), dtype=torch.float64, device=scores.device),
                        torch.zeros(neg.numel(), dtype=torch.float64, device=scores.device)])
    order = torch.argsort(scores, descending=True, stable=True)
    s, y = scores[order], labels[order]
    tp = torch.cumsum(y, 0)
    k = torch.arange(1, s.numel() + 1, dtype=torch.float64, device=s.device)
    last_of_group = torch.ones_like(s, dtype=torch.bool)
    last_of_group[:-1] = s[1:] != s[:-1]
    tp_g, k_g = tp[last_of_group], k[last_of_group]
    precision = tp_g / k_g
    recall = tp_g / max(pos.numel(), 1)
    prev = torch.cat([torch.zeros(1, dtype=torch.float64, device=s.device), recall[:-1]])
    return ((recall - prev) * precision).sum()


# ---------------------------------------------------------------------------------------------------
# reduction
# ---------------------------------------------------------------------------------------------------
def reduce_metric_sums(local_sums: torch.Tensor, group=None) -> torch.Tensor:
    """All-reduce (sum) of a small float64 vector of per-rank metric sums, e.g. [sum AP, sum AUC, #batches].
    No-op when torch.distributed is not initialised (single GPU)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(local_sums, op=dist.ReduceOp.SUM, group=group)
    return local_sums


def evaluate_sharded(step_fn: Callable[[int], Tuple[torch.Tensor, torch.Tensor]], num_batches: int,
                     rank: int = 0, world_size: int = 1, device: Optional[torch.device] = None, group=None) -> dict:
    """The evaluation loop of evaluate_models_utils.py:49-152, sharded: `step_fn(batch_index)` returns the
    positive and negative probabilities of one batch (evaluate_models_utils.py:140-141); each rank runs its
    own batches, accumulates [sum AP, sum AUC, count] on its device and ONE all-reduce at the end gives every
    rank the global means (the reference averages the per-batch metrics, train_link_prediction.py:301-306)."""
    sums = None
    for i in shard_batch_indices(num_batches, rank, world_size):
        pos, neg = step_fn(i)
        m = torch.stack([average_precision(pos, neg), binary_auc(pos, neg),
                         torch.ones((), dtype=torch.float64, device=pos.device)])
        sums = m if sums is None else sums + m
    if sums is None:       # a rank without batches still takes part in the collective
        sums = torch.zeros(3, dtype=torch.float64, device=device or "cpu")
    sums = reduce_metric_sums(sums, group)
    n = float(sums[2].item())
    return {"average_precision": float(sums[0].item()) / max(n, 1.0), "roc_auc": float(sums[1].item()) / max(n, 1.0),
            "num_batches": int(n)}


# ---------------------------------------------------------------------------------------------------
# data-parallel training (last paragraph): every rank runs its own 200-edge batch through the
# training forward / backward, then ONE flat all-reduce averages the gradients (1.05 M fp32 = 4.2 MB for
# DyGFormer + 0.06 M for the MergeLayer: a single bucket — xGMI rings are per-link bound at ~153 GB/s, so one
# 4 MB ring all-reduce costs ~50 us + latency; splitting it into per-tensor collectives would only add latency).
# The effective batch becomes world * 200, as with torch DDP.
# ---------------------------------------------------------------------------------------------------
def allreduce_gradients(parameters: Iterable[torch.nn.Parameter], group=None) -> int:
    """Average the .grad of `parameters` over the ranks in one flattened bucket (parameters without a gradient take part
    with zeros, so every rank issues the same collective).  Returns the bucket's element count.  No-op for one rank."""
    params = [p for p in parameters if p.requires_grad]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1 or not params:
        return 0
    world = dist.get_world_size(group)
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= world
    o = 0
    for p in params:
        n = p.numel()
        g = flat[o:o + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        o += n
    return int(flat.numel())
