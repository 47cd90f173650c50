"""Synthetic temporal-interaction graphs and DyGFormer parameter sets.

No dataset file ships with the reference (processed_data/ is empty and the
Zenodo download of README.md:25-27 is unreachable offline), so the tests, the
golden-vector generator and bench.py all drive the hot path with graphs in the
reference's in-memory format: the five parallel arrays of `Data`
(utils/DataLoader.py:46-64) plus the two feature tables whose row 0 is the
all-zero padding row (preprocess_data/preprocess_data.py:101-108).

Everything is derived from `numpy.random.RandomState(seed)` (the legacy,
version-stable stream), so only *outputs* have to be stored in fixtures.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np

NODE_FEAT_DIM = 172  # utils/DataLoader.py:84-90 pads both tables to 172


@dataclass
class InteractionData:
    """Same fields as the reference `Data` container (utils/DataLoader.py:46-64)."""
    src_node_ids: np.ndarray      # int64 [E]
    dst_node_ids: np.ndarray      # int64 [E]
    node_interact_times: np.ndarray  # float64 [E]
    edge_ids: np.ndarray          # int64 [E]
    labels: np.ndarray            # float64 [E]

    @property
    def num_interactions(self) -> int:
        return len(self.src_node_ids)

    @property
    def max_node_id(self) -> int:
        return int(max(self.src_node_ids.max(), self.dst_node_ids.max()))


def zipf_choice(rs: np.random.RandomState, n_items: int, size: int, a: float) -> np.ndarray:
    ranks = np.arange(1, n_items + 1, dtype=np.float64)
    p = ranks ** (-a)
    p /= p.sum()
    return rs.choice(n_items, size=size, p=p)


def make_bipartite_graph(num_users: int, num_items: int, num_edges: int, seed: int = 0,
                         zipf_a: float = 0.9, time_span: float = 2.68e6,
                         edge_feat_dim: int = NODE_FEAT_DIM, edge_feat_kind: str = "normal",
                         duplicate_time_every: int = 0
                         ) -> Tuple[InteractionData, np.ndarray, np.ndarray]:
    """SURVEY.md §8(d) generator: users 1..U, items U+1..U+I, edge ids 1..E,
    chronological float64 timestamps, Zipf endpoint popularity.

    `duplicate_time_every` > 0 forces every n-th timestamp to equal its
    predecessor, to exercise the stable tie order (utils/utils.py:98-100) and
    the strictly-earlier rule (utils/utils.py:139-141).
    Returns (data, node_raw_features[N+1,172], edge_raw_features[E+1,F_e]).
    """
    rs = np.random.RandomState(seed)
    src = zipf_choice(rs, num_users, num_edges, zipf_a).astype(np.int64) + 1
    dst = zipf_choice(rs, num_items, num_edges, zipf_a).astype(np.int64) + 1 + num_users
    ts = np.sort(rs.uniform(0.0, time_span, size=num_edges)).astype(np.float64)
    if duplicate_time_every > 0:
        idx = np.arange(duplicate_time_every, num_edges, duplicate_time_every)
        ts[idx] = ts[idx - 1]
    eid = np.arange(1, num_edges + 1, dtype=np.int64)
    labels = np.zeros(num_edges, dtype=np.float64)
    n_nodes = num_users + num_items
    node_feat = np.zeros((n_nodes + 1, NODE_FEAT_DIM), dtype=np.float32)
    edge_feat = np.zeros((num_edges + 1, edge_feat_dim), dtype=np.float32)
    if edge_feat_kind == "normal":
        edge_feat[1:] = rs.standard_normal((num_edges, edge_feat_dim)).astype(np.float32)
    elif edge_feat_kind == "sparse4":          # MOOC-like: 4 random + rest zero columns
        edge_feat[1:, :4] = rs.standard_normal((num_edges, 4)).astype(np.float32)
    elif edge_feat_kind == "zeros":            # LastFM-like
        pass
    else:
        raise ValueError(f"unknown edge_feat_kind {edge_feat_kind}")
    data = InteractionData(src, dst, ts, eid, labels)
    return data, node_feat, edge_feat


def make_general_graph(num_nodes: int, num_edges: int, seed: int = 0, time_span: float = 1000.0,
                       integer_times: bool = True, node_feat_kind: str = "normal"
                       ) -> Tuple[InteractionData, np.ndarray, np.ndarray]:
    """Small non-bipartite graph for edge-case fixtures: repeated (src,dst) pairs,
    self-interactions, many duplicate (integer) timestamps, non-zero node features,
    NOT necessarily chronological edge order (exercises the stable per-node sort)."""
    rs = np.random.RandomState(seed)
    src = rs.randint(1, num_nodes + 1, size=num_edges).astype(np.int64)
    dst = rs.randint(1, num_nodes + 1, size=num_edges).astype(np.int64)
    ts = rs.uniform(0.0, time_span, size=num_edges)
    if integer_times:
        ts = np.floor(ts)
    ts = ts.astype(np.float64)
    # mostly chronological, with a shuffled tail so per-node stable sorting matters
    order = np.argsort(ts, kind="stable")
    tail = num_edges // 5
    order[-tail:] = order[-tail:][rs.permutation(tail)]
    src, dst, ts = src[order], dst[order], ts[order]
    eid = np.arange(1, num_edges + 1, dtype=np.int64)
    labels = np.zeros(num_edges, dtype=np.float64)
    node_feat = np.zeros((num_nodes + 1, NODE_FEAT_DIM), dtype=np.float32)
    if node_feat_kind == "normal":
        node_feat[1:] = rs.standard_normal((num_nodes, NODE_FEAT_DIM)).astype(np.float32)
    edge_feat = np.zeros((num_edges + 1, NODE_FEAT_DIM), dtype=np.float32)
    edge_feat[1:] = rs.standard_normal((num_edges, NODE_FEAT_DIM)).astype(np.float32)
    return InteractionData(src, dst, ts, eid, labels), node_feat, edge_feat


# --------------------------------------------------------------------------------------
# Parameter sets.  Keys and shapes are the reference state_dict (SURVEY.md Appendix A,
# models/DyGFormer.py:47-66, :332-335, :429-440; models/modules.py:19-21, :53-54).
# --------------------------------------------------------------------------------------

def dygformer_param_shapes(node_feat_dim: int, edge_feat_dim: int, time_feat_dim: int,
                           channel_embedding_dim: int, patch_size: int, num_layers: int
                           ) -> Dict[str, Tuple[int, ...]]:
    C = channel_embedding_dim
    D = 4 * C
    P = patch_size
    shapes: Dict[str, Tuple[int, ...]] = {
        "time_encoder.w.weight": (time_feat_dim, 1),
        "time_encoder.w.bias": (time_feat_dim,),
        "neighbor_co_occurrence_encoder.neighbor_co_occurrence_encode_layer.0.weight": (C, 1),
        "neighbor_co_occurrence_encoder.neighbor_co_occurrence_encode_layer.0.bias": (C,),
        "neighbor_co_occurrence_encoder.neighbor_co_occurrence_encode_layer.2.weight": (C, C),
        "neighbor_co_occurrence_encoder.neighbor_co_occurrence_encode_layer.2.bias": (C,),
        "projection_layer.node.weight": (C, P * node_feat_dim),
        "projection_layer.node.bias": (C,),
        "projection_layer.edge.weight": (C, P * edge_feat_dim),
        "projection_layer.edge.bias": (C,),
        "projection_layer.time.weight": (C, P * time_feat_dim),
        "projection_layer.time.bias": (C,),
        "projection_layer.neighbor_co_occurrence.weight": (C, P * C),
        "projection_layer.neighbor_co_occurrence.bias": (C,),
    }
    for l in range(num_layers):
        p = f"transformers.{l}."
        shapes[p + "multi_head_attention.in_proj_weight"] = (3 * D, D)
        shapes[p + "multi_head_attention.in_proj_bias"] = (3 * D,)
        shapes[p + "multi_head_attention.out_proj.weight"] = (D, D)
        shapes[p + "multi_head_attention.out_proj.bias"] = (D,)
        shapes[p + "linear_layers.0.weight"] = (4 * D, D)
        shapes[p + "linear_layers.0.bias"] = (4 * D,)
        shapes[p + "linear_layers.1.weight"] = (D, 4 * D)
        shapes[p + "linear_layers.1.bias"] = (D,)
        shapes[p + "norm_layers.0.weight"] = (D,)
        shapes[p + "norm_layers.0.bias"] = (D,)
        shapes[p + "norm_layers.1.weight"] = (D,)
        shapes[p + "norm_layers.1.bias"] = (D,)
    shapes["output_layer.weight"] = (node_feat_dim, D)
    shapes["output_layer.bias"] = (node_feat_dim,)
    return shapes


def make_dygformer_params(seed: int, node_feat_dim: int = NODE_FEAT_DIM, edge_feat_dim: int = NODE_FEAT_DIM,
                          time_feat_dim: int = 100, channel_embedding_dim: int = 50, patch_size: int = 1,
                          num_layers: int = 2) -> Dict[str, np.ndarray]:
    """Deterministic float32 parameters with PyTorch-default-like magnitudes
    (uniform +-1/sqrt(fan_in) for Linear, 1+-0.1 / +-0.1 for LayerNorm, and the
    reference's 10^-linspace(0,9) time frequencies, models/modules.py:20, with a
    small perturbation and a non-zero bias so the trainable path is exercised)."""
    rs = np.random.RandomState(seed)
    out: Dict[str, np.ndarray] = {}
    shapes = dygformer_param_shapes(node_feat_dim, edge_feat_dim, time_feat_dim,
                                    channel_embedding_dim, patch_size, num_layers)
    for key, shape in shapes.items():
        if key == "time_encoder.w.weight":
            base = (1.0 / 10 ** np.linspace(0, 9, time_feat_dim, dtype=np.float32)).reshape(shape)
            val = base * (1.0 + 0.01 * rs.uniform(-1, 1, size=shape))
        elif key == "time_encoder.w.bias":
            val = 0.1 * rs.uniform(-1, 1, size=shape)
        elif "norm_layers" in key and key.endswith("weight"):
            val = 1.0 + 0.1 * rs.uniform(-1, 1, size=shape)
        elif "norm_layers" in key:
            val = 0.1 * rs.uniform(-1, 1, size=shape)
        else:
            fan_in = shape[1] if len(shape) == 2 else None
            if fan_in is None:
                # bias of the Linear whose weight was generated just before ("...bias" -> "...weight")
                fan_in = out[key[:-4] + "weight"].shape[1]
            bound = 1.0 / np.sqrt(fan_in)
            val = rs.uniform(-bound, bound, size=shape)
        out[key] = np.ascontiguousarray(val, dtype=np.float32)
    return out


def make_merge_layer_params(seed: int, dim: int = NODE_FEAT_DIM) -> Dict[str, np.ndarray]:
    """MergeLayer(172,172,172,1) link predictor (models/modules.py:42-68)."""
    rs = np.random.RandomState(seed)
    b1 = 1.0 / np.sqrt(2 * dim)
    b2 = 1.0 / np.sqrt(dim)
    return {
        "fc1.weight": rs.uniform(-b1, b1, size=(dim, 2 * dim)).astype(np.float32),
        "fc1.bias": rs.uniform(-b1, b1, size=(dim,)).astype(np.float32),
        "fc2.weight": rs.uniform(-b2, b2, size=(1, dim)).astype(np.float32),
        "fc2.bias": rs.uniform(-b2, b2, size=(1,)).astype(np.float32),
    }


def random_negative_dst(rs: np.random.RandomState, unique_dst: np.ndarray, size: int) -> np.ndarray:
    """`random` negative sampling with a seed: destinations drawn uniformly from the
    unique destination ids (utils/utils.py NegativeEdgeSampler.random_sample)."""
    idx = rs.randint(0, len(unique_dst), size)
    return unique_dst[idx].astype(np.int64)


def tgat_param_shapes(node_feat_dim: int = NODE_FEAT_DIM, edge_feat_dim: int = NODE_FEAT_DIM, time_feat_dim: int = 100,
                      num_layers: int = 2) -> Dict[str, Tuple[int, ...]]:
    """state_dict of the reference TGAT (models/TGAT.py:34-45, models/modules.py:121-133, :53-54)."""
    Dq, Dkv = node_feat_dim + time_feat_dim, node_feat_dim + edge_feat_dim + time_feat_dim
    shapes: Dict[str, Tuple[int, ...]] = {"time_encoder.w.weight": (time_feat_dim, 1), "time_encoder.w.bias": (time_feat_dim,)}
    for l in range(num_layers):
        p = f"temporal_conv_layers.{l}."
        shapes[p + "query_projection.weight"] = (Dq, Dq)
        shapes[p + "key_projection.weight"] = (Dq, Dkv)
        shapes[p + "value_projection.weight"] = (Dq, Dkv)
        shapes[p + "layer_norm.weight"] = (Dq,)
        shapes[p + "layer_norm.bias"] = (Dq,)
        shapes[p + "residual_fc.weight"] = (Dq, Dq)
        shapes[p + "residual_fc.bias"] = (Dq,)
    for l in range(num_layers):
        p = f"merge_layers.{l}."
        shapes[p + "fc1.weight"] = (node_feat_dim, Dq + node_feat_dim)
        shapes[p + "fc1.bias"] = (node_feat_dim,)
        shapes[p + "fc2.weight"] = (node_feat_dim, node_feat_dim)
        shapes[p + "fc2.bias"] = (node_feat_dim,)
    return shapes


def make_tgat_params(seed: int, node_feat_dim: int = NODE_FEAT_DIM, edge_feat_dim: int = NODE_FEAT_DIM, time_feat_dim: int = 100,
                     num_layers: int = 2) -> Dict[str, np.ndarray]:
    rs = np.random.RandomState(seed)
    out: Dict[str, np.ndarray] = {}
    for key, shape in tgat_param_shapes(node_feat_dim, edge_feat_dim, time_feat_dim, num_layers).items():
        if key == "time_encoder.w.weight":
            base = (1.0 / 10 ** np.linspace(0, 9, time_feat_dim, dtype=np.float32)).reshape(shape)
            val = base * (1.0 + 0.01 * rs.uniform(-1, 1, size=shape))
        elif key == "time_encoder.w.bias":
            val = 0.1 * rs.uniform(-1, 1, size=shape)
        elif "layer_norm" in key:
            val = (1.0 if key.endswith("weight") else 0.0) + 0.1 * rs.uniform(-1, 1, size=shape)
        else:
            fan_in = shape[1] if len(shape) == 2 else out[key[:-4] + "weight"].shape[1]
            val = rs.uniform(-1, 1, size=shape) / np.sqrt(fan_in)
        out[key] = np.ascontiguousarray(val, dtype=np.float32)
    return out


def tgn_param_shapes(num_nodes: int, node_feat_dim: int = NODE_FEAT_DIM, edge_feat_dim: int = NODE_FEAT_DIM, time_feat_dim: int = 100,
                     num_layers: int = 1) -> Dict[str, Tuple[int, ...]]:
    """state_dict of the reference MemoryModel('TGN') (models/MemoryModel.py:43-74): the memory bank is registered twice
    (memory_bank.* and memory_updater.memory_bank.*, same tensors)."""
    Fn = node_feat_dim
    Dm = 2 * Fn + time_feat_dim + edge_feat_dim
    shapes: Dict[str, Tuple[int, ...]] = {"time_encoder.w.weight": (time_feat_dim, 1), "time_encoder.w.bias": (time_feat_dim,),
                                          "memory_bank.node_memories": (num_nodes, Fn), "memory_bank.node_last_updated_times": (num_nodes,),
                                          "memory_updater.memory_bank.node_memories": (num_nodes, Fn),
                                          "memory_updater.memory_bank.node_last_updated_times": (num_nodes,),
                                          "memory_updater.memory_updater.weight_ih": (3 * Fn, Dm), "memory_updater.memory_updater.weight_hh": (3 * Fn, Fn),
                                          "memory_updater.memory_updater.bias_ih": (3 * Fn,), "memory_updater.memory_updater.bias_hh": (3 * Fn,)}
    for k, v in tgat_param_shapes(node_feat_dim, edge_feat_dim, time_feat_dim, num_layers).items():
        shapes["embedding_module." + k] = v          # includes embedding_module.time_encoder.* (the shared module)
    return shapes


def make_tgn_params(seed: int, num_nodes: int, num_layers: int = 1) -> Dict[str, np.ndarray]:
    """Trainable TGN parameters (the memory bank starts at zero and is not part of this set)."""
    rs = np.random.RandomState(seed)
    out = {("embedding_module." + k if not k.startswith("time_encoder") else k): v for k, v in make_tgat_params(seed, num_layers=num_layers).items()}
    for k in ("time_encoder.w.weight", "time_encoder.w.bias"):
        out["embedding_module." + k] = out[k]        # shared module: both names, same values
    Fn, Dm = NODE_FEAT_DIM, 2 * NODE_FEAT_DIM + 100 + NODE_FEAT_DIM
    b = 1.0 / np.sqrt(Fn)
    out["memory_updater.memory_updater.weight_ih"] = rs.uniform(-b, b, (3 * Fn, Dm)).astype(np.float32)
    out["memory_updater.memory_updater.weight_hh"] = rs.uniform(-b, b, (3 * Fn, Fn)).astype(np.float32)
    out["memory_updater.memory_updater.bias_ih"] = rs.uniform(-b, b, (3 * Fn,)).astype(np.float32)
    out["memory_updater.memory_updater.bias_hh"] = rs.uniform(-b, b, (3 * Fn,)).astype(np.float32)
    return out
