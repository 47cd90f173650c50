"""The evaluation loop around the hot path (reference evaluate_models_utils.py:18-153) and its negative edge sampler
(utils/utils.py:305-495), with the reference's names and argument meaning.  What changes is where the work happens:
the scores of a batch never leave the GPU (MergeLayer + sigmoid, BCELoss, average precision and ROC AUC are HIP
kernels, dyglib_amd/csrc/metrics.hip), DyGFormer batches are grouped `fuse_batches` at a time into one launch, and the
host synchronises once, at the end, instead of once per batch."""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch
import torch.nn as nn

from .metrics import link_prediction_metrics_device


class NegativeEdgeSampler(object):
    """utils/utils.py:305-495, strategy 'random' (the reference's default, train_link_prediction.py:95-104): sources and
    destinations drawn independently from the unique source / destination ids with the sampler's own RandomState, so a
    seeded sampler returns the reference's negatives draw for draw.  'historical' / 'inductive' enumerate Python sets of
    edge tuples, whose iteration order decides the result; they are not restated here and raise NotImplementedError."""

    def __init__(self, src_node_ids: np.ndarray, dst_node_ids: np.ndarray, interact_times: np.ndarray = None,
                 last_observed_time: float = None, negative_sample_strategy: str = "random", seed: int = None):
        self.seed = seed
        self.negative_sample_strategy = negative_sample_strategy
        self.src_node_ids = src_node_ids
        self.dst_node_ids = dst_node_ids
        self.interact_times = interact_times
        self.unique_src_node_ids = np.unique(src_node_ids)
        self.unique_dst_node_ids = np.unique(dst_node_ids)
        self.last_observed_time = last_observed_time
        if negative_sample_strategy in ("historical", "inductive"):
            raise NotImplementedError(f"negative_sample_strategy {negative_sample_strategy} is not available (only 'random')")
        if negative_sample_strategy != "random":
            raise ValueError(f"Not implemented error for negative_sample_strategy {negative_sample_strategy}!")   # utils/utils.py:375
        if self.seed is not None:
            self.random_state = np.random.RandomState(self.seed)

    def sample(self, size: int, batch_src_node_ids: np.ndarray = None, batch_dst_node_ids: np.ndarray = None,
               current_batch_start_time: float = 0.0, current_batch_end_time: float = 0.0):
        return self.random_sample(size=size)

    def random_sample(self, size: int):
        rs = np.random if self.seed is None else self.random_state                 # utils/utils.py:384-389
        src_idx = rs.randint(0, len(self.unique_src_node_ids), size)
        dst_idx = rs.randint(0, len(self.unique_dst_node_ids), size)
        return self.unique_src_node_ids[src_idx], self.unique_dst_node_ids[dst_idx]

    def reset_random_state(self):
        self.random_state = np.random.RandomState(self.seed)


def get_idx_data_loader(indices_list: list, batch_size: int, shuffle: bool):
    """utils/DataLoader.py:29-43: batches of indices, last batch kept (drop_last=False)."""
    from torch.utils.data import DataLoader
    return DataLoader(dataset=list(indices_list), batch_size=batch_size, shuffle=shuffle, drop_last=False)


def _is_plain_bce(loss_func) -> bool:
    return isinstance(loss_func, nn.BCELoss) and loss_func.reduction == "mean" and getattr(loss_func, "weight", None) is None


def evaluate_model_link_prediction(model_name: str, model: nn.Module, neighbor_sampler, evaluate_idx_data_loader,
                                   evaluate_neg_edge_sampler: NegativeEdgeSampler, evaluate_data, loss_func: nn.Module,
                                   num_neighbors: int = 20, time_gap: int = 2000, fuse_batches: int = 32, tgat_fuse_batches: int = 128
                                   ) -> Tuple[List[float], List[dict]]:
    """evaluate_models_utils.py:18-153 for the models of this package (DyGFormer, TGAT, TGN): `model` is
    nn.Sequential(backbone, MergeLayer); returns (evaluate_losses, evaluate_metrics) = one float and one
    {'average_precision', 'roc_auc'} dict per batch, like the reference.
    `fuse_batches` batches form one launch; TGAT with `recent` sampling takes `tgat_fuse_batches`: its rows do not depend on the batch they are in,
    and the more batches a call holds the more of its level-1 (node, time) entries repeat and are computed once (128 batches per call: 1.4x the
    rate of 32; ~23 GB of level arrays and layer buffers at the Reddit shape — lower it on a smaller device)."""
    assert evaluate_neg_edge_sampler.seed is not None                           # evaluate_models_utils.py:35
    evaluate_neg_edge_sampler.reset_random_state()
    if model_name not in ("DyGFormer", "TGAT", "TGN"):
        raise ValueError(f"Wrong value for model_name {model_name}!")
    model[0].set_neighbor_sampler(neighbor_sampler)
    model.eval()
    backbone, merge = model[0], model[1]
    plain_bce = _is_plain_bce(loss_func)
    results = []                                    # per launch: (ap, auc, loss, status) device tensors, in batch order
    label_cache = {}

    def score(groups_pos, groups_neg):
        """groups_*: lists of equally sized (src, dst, t[, eid]) tuples -> predicts, labels [len, 2B]"""
        n, B = len(groups_pos), len(groups_pos[0][0])
        if model_name == "DyGFormer":
            src = np.stack([g[0] for g in groups_pos] + [g[0] for g in groups_neg])
            dst = np.stack([g[1] for g in groups_pos] + [g[1] for g in groups_neg])
            tms = np.stack([g[2] for g in groups_pos] + [g[2] for g in groups_neg])
            a, b = backbone.compute_src_dst_node_temporal_embeddings_many(src, dst, tms, pos_neg_halves=True)      # [positives ; negatives]
            prob = merge.link_probabilities(a.reshape(2 * n * B, -1), b.reshape(2 * n * B, -1)).reshape(2, n, B)
        elif model_name == "TGAT" and neighbor_sampler.sample_neighbor_strategy == "recent":
            # rows do not depend on the batch they are in (fixed k, stateless sampling): the n batches are one call on n*B edges
            # positives and negatives in ONE call: level de-duplication (tgat.hip) then computes the shared source side once
            # positives and negatives in ONE call on the roots [sources ; destinations ; negative destinations]: the negative call's sources are
            # the positive call's (evaluate_models_utils.py:62-63), and level de-duplication (tgat.hip) computes every other repeated entry once
            catp = lambda c: np.concatenate([g[c] for g in groups_pos])
            se, de, ne = backbone.compute_step_embeddings(catp(0), catp(1), np.concatenate([g[1] for g in groups_neg]), catp(2), num_neighbors=num_neighbors)
            prob = merge.link_probabilities(torch.cat([se, se]), torch.cat([de, ne])).reshape(2, n, B)
        elif model_name == "TGAT":      # random strategies consume the sampler's RandomState call by call: keep the reference's call order
            probs = []
            for gp, gn in zip(groups_pos, groups_neg):
                pe = backbone.compute_src_dst_node_temporal_embeddings(gp[0], gp[1], gp[2], num_neighbors=num_neighbors)
                ne = backbone.compute_src_dst_node_temporal_embeddings(gn[0], gn[1], gn[2], num_neighbors=num_neighbors)
                probs.append(torch.stack([merge.link_probabilities(*pe), merge.link_probabilities(*ne)]))
            prob = torch.stack(probs, dim=1)
        else:       # TGN: batches strictly in sequence; the negative and the positive call of a batch (:85-107) are one library call
            probs = []
            for gp, gn in zip(groups_pos, groups_neg):
                if all(isinstance(x, np.ndarray) for x in (gp[0], gp[1], gn[0], gn[1], gp[2])) and len(gp[0]) == len(gn[0]):
                    # host batches: [positives ; negatives] joined on the host, one library call, one link-predictor launch over its output
                    n = len(gp[0])
                    se, de = backbone.compute_step_embeddings_joint(np.concatenate([gp[0], gn[0]]), np.concatenate([gp[1], gn[1]]), np.concatenate([gp[2], gp[2]]),
                                                                    gp[3], n, num_neighbors=num_neighbors)
                    probs.append(merge.link_probabilities(se, de).view(2, n))
                    continue
                ps, pd, ns, nd = backbone.compute_step_embeddings(gp[0], gp[1], gn[0], gn[1], gp[2], gp[3], num_neighbors=num_neighbors)
                probs.append(torch.stack([merge.link_probabilities(ps, pd), merge.link_probabilities(ns, nd)]))
            prob = torch.stack(probs, dim=1)
        G, Bp = prob.shape[1], prob.shape[2]
        predicts = prob.permute(1, 0, 2).reshape(G, 2 * Bp)                                 # :142 cat([positive, negative]) per batch: a view for one batch per launch
        key = (G, Bp, prob.device, prob.dtype)
        labels = label_cache.get(key)                                                       # :143 [ones ; zeros]: the same tensor for every launch of a shape
        if labels is None:
            labels = label_cache[key] = torch.cat([torch.ones((G, Bp), device=prob.device, dtype=prob.dtype),
                                                   torch.zeros((G, Bp), device=prob.device, dtype=prob.dtype)], dim=1)
        return predicts, labels

    def flush(pos, neg):
        if not pos:
            return
        predicts, labels = score(pos, neg)
        ap, auc, loss, status = link_prediction_metrics_device(predicts, labels)
        if not plain_bce:
            loss = torch.stack([loss_func(input=predicts[i], target=labels[i]).double() for i in range(len(pos))])
        results.append((ap, auc, loss, status))

    group_limit = max(1, tgat_fuse_batches if (model_name == "TGAT" and neighbor_sampler.sample_neighbor_strategy == "recent") else fuse_batches)
    with torch.no_grad():
        pend_pos, pend_neg = [], []
        for evaluate_data_indices in evaluate_idx_data_loader:
            idx = evaluate_data_indices.numpy() if isinstance(evaluate_data_indices, torch.Tensor) else np.asarray(evaluate_data_indices)
            src, dst = evaluate_data.src_node_ids[idx], evaluate_data.dst_node_ids[idx]
            tms, eid = evaluate_data.node_interact_times[idx], evaluate_data.edge_ids[idx]
            _, neg_dst = evaluate_neg_edge_sampler.sample(size=len(src))                    # :64-66 ('random')
            if pend_pos and (len(pend_pos[0][0]) != len(src) or len(pend_pos) >= group_limit):
                flush(pend_pos, pend_neg)
                pend_pos, pend_neg = [], []
            pend_pos.append((src, dst, tms, eid))
            pend_neg.append((src, neg_dst, tms, None))
        flush(pend_pos, pend_neg)

    if not results:
        return [], []
    ap, auc, loss, status = (torch.cat([r[i] for r in results]).cpu().numpy() for i in range(4))     # the only synchronisation
    if status.any():
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    return [float(v) for v in loss], [{"average_precision": float(a), "roc_auc": float(u)} for a, u in zip(ap, auc)]
