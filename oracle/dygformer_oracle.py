"""CPU ORACLE — test infrastructure, NOT product code.

A CPU restatement (numpy for the integer/index work, PyTorch-CPU fp32 functional
ops for the dense math) of the reference's temporal-neighbour-aggregation path:

  * NeighborSampler construction / find_neighbors_before / get_historical_neighbors
    (`recent`) / get_all_first_hop_neighbors      -> /root/reference/utils/utils.py:71-302
  * DyGFormer.compute_src_dst_node_temporal_embeddings and everything under it
                                                  -> /root/reference/models/DyGFormer.py:68-461
  * TimeEncoder, MergeLayer                       -> /root/reference/models/modules.py:7-68

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module, and only as the checker / the reported CPU baseline.  Nothing
under `dyglib_amd/` imports it; the product path fails loudly without its HIP
library instead of falling back to this code.

Parity status: PINNED — `tests/test_oracle_golden.py` checks every function here
against golden vectors produced by importing the reference itself in the build
container (`oracle/make_golden.py`, outputs under `tests/golden/`).

Each function cites the reference file:line it restates.  The code is written
per row where the reference is per row (searchsorted, np.unique) so that, timed
as `cpu_baseline` (kind "port"), it has the reference's cost profile rather than
that of an optimised variant.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# ======================================================================================
# L1: time-sorted adjacency  (utils/utils.py:283-302 builder, :73-110 constructor)
# ======================================================================================

class OracleAdjacency:
    """Per-node neighbour id / edge id / timestamp arrays, ascending in time with ties
    kept in edge-list order (stable sort, utils/utils.py:98-100).  Node 0 is the padding
    node with an empty row (utils/utils.py:94-97)."""

    def __init__(self, src: np.ndarray, dst: np.ndarray, eid: np.ndarray, ts: np.ndarray):
        max_node = int(max(src.max(), dst.max()))                      # utils.py:293
        n = max_node + 1
        # undirected: every interaction is appended to both endpoints, src entry first
        # (utils.py:298-300), in edge-list order.
        owner = np.empty(2 * len(src), dtype=np.int64)
        other = np.empty(2 * len(src), dtype=np.int64)
        owner[0::2], owner[1::2] = src, dst
        other[0::2], other[1::2] = dst, src
        e2 = np.repeat(np.asarray(eid, dtype=np.int64), 2)
        t2 = np.repeat(np.asarray(ts, dtype=np.float64), 2)
        # stable by (owner) keeps edge-list order inside a node; then stable by time
        # inside the node == sorted(key=ts) of utils.py:100.
        order = np.lexsort((np.arange(len(owner)), t2, owner))
        owner, other, e2, t2 = owner[order], other[order], e2[order], t2[order]
        counts = np.bincount(owner, minlength=n)
        self.indptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(counts, out=self.indptr[1:])
        self.nbr = other
        self.eid = e2
        self.ts = t2
        self.num_nodes = n

    def row(self, node: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        a, b = self.indptr[node], self.indptr[node + 1]
        return self.nbr[a:b], self.eid[a:b], self.ts[a:b]


def find_neighbors_before(adj: OracleAdjacency, node: int, t: float):
    """utils/utils.py:130-147 — prefix of interactions strictly earlier than t
    (np.searchsorted default side='left')."""
    nbr, eid, ts = adj.row(int(node))
    i = int(np.searchsorted(ts, t))                                    # utils.py:141
    return nbr[:i], eid[:i], ts[:i]


def get_all_first_hop_neighbors(adj: OracleAdjacency, node_ids: np.ndarray, times: np.ndarray):
    """utils/utils.py:254-273."""
    ids_l, eids_l, ts_l = [], [], []
    for node, t in zip(node_ids, times):
        a, b, c = find_neighbors_before(adj, node, t)
        ids_l.append(a), eids_l.append(b), ts_l.append(c)
    return ids_l, eids_l, ts_l


def get_historical_neighbors_recent(adj: OracleAdjacency, node_ids: np.ndarray, times: np.ndarray, k: int):
    """utils/utils.py:149-214, `recent` branch (:200-209): most recent k, right-aligned,
    zero filled; ids/eids int64, times stored as float32 (:161-167)."""
    assert k > 0, "Number of sampled neighbors for each node should be greater than 0!"  # utils.py:157
    B = len(node_ids)
    out_n = np.zeros((B, k), dtype=np.int64)
    out_e = np.zeros((B, k), dtype=np.int64)
    out_t = np.zeros((B, k), dtype=np.float32)
    for r, (node, t) in enumerate(zip(node_ids, times)):
        nbr, eid, ts = find_neighbors_before(adj, node, t)
        if len(nbr) > 0:
            nbr, eid, ts = nbr[-k:], eid[-k:], ts[-k:]
            out_n[r, k - len(nbr):] = nbr
            out_e[r, k - len(eid):] = eid
            out_t[r, k - len(ts):] = ts
    return out_n, out_e, out_t


# ======================================================================================
# DyGFormer host-side sequence building  (models/DyGFormer.py:196-245, :337-393)
# ======================================================================================

def pad_sequences(node_ids: np.ndarray, times: np.ndarray, ids_l: List[np.ndarray], eids_l: List[np.ndarray],
                  ts_l: List[np.ndarray], patch_size: int, max_input_sequence_length: int):
    """models/DyGFormer.py:196-245: keep the most recent L-1 interactions (:214-218); S = batch
    max length + 1, rounded up to a multiple of the patch size (:223-226); column 0 is the target
    node itself (id, edge 0, query time) and the history follows oldest->newest (:234-242)."""
    L = max_input_sequence_length
    assert L - 1 > 0, "Maximal number of neighbors for each node should be greater than 1!"  # :209
    ids_l = [x[-(L - 1):] if len(x) > L - 1 else x for x in ids_l]
    eids_l = [x[-(L - 1):] if len(x) > L - 1 else x for x in eids_l]
    ts_l = [x[-(L - 1):] if len(x) > L - 1 else x for x in ts_l]
    S = max([len(x) for x in ids_l], default=0) + 1
    if S % patch_size != 0:
        S += patch_size - S % patch_size
    B = len(node_ids)
    pid = np.zeros((B, S), dtype=np.int64)
    pe = np.zeros((B, S), dtype=np.int64)
    pt = np.zeros((B, S), dtype=np.float32)
    for r in range(B):
        pid[r, 0] = node_ids[r]
        pt[r, 0] = times[r]
        n = len(ids_l[r])
        if n > 0:
            pid[r, 1:n + 1] = ids_l[r]
            pe[r, 1:n + 1] = eids_l[r]
            pt[r, 1:n + 1] = ts_l[r]
    return pid, pe, pt


def first_hop_windows(adj: OracleAdjacency, node_ids, times, patch_size: int, max_input_sequence_length: int):
    """get_all_first_hop_neighbors + pad_sequences, i.e. models/DyGFormer.py:78-100 for one side."""
    a, b, c = get_all_first_hop_neighbors(adj, node_ids, times)
    return pad_sequences(node_ids, times, a, b, c, patch_size, max_input_sequence_length)


def count_nodes_appearances(src_ids: np.ndarray, dst_ids: np.ndarray):
    """models/DyGFormer.py:337-393.  For every position of the src row: [count in src row,
    count in dst row] (:374); for every position of the dst row: [count in src row, count in
    dst row] (:380).  Positions holding the padding id 0 are zeroed (:389-391).  float32."""
    B = src_ids.shape[0]
    src_out = np.zeros(src_ids.shape + (2,), dtype=np.float32)
    dst_out = np.zeros(dst_ids.shape + (2,), dtype=np.float32)
    for r in range(B):
        s_keys, s_inv, s_cnt = np.unique(src_ids[r], return_inverse=True, return_counts=True)   # :354
        d_keys, d_inv, d_cnt = np.unique(dst_ids[r], return_inverse=True, return_counts=True)   # :364
        s_map = dict(zip(s_keys.tolist(), s_cnt.tolist()))
        d_map = dict(zip(d_keys.tolist(), d_cnt.tolist()))
        src_out[r, :, 0] = s_cnt[s_inv]
        src_out[r, :, 1] = [d_map.get(v, 0) for v in src_ids[r].tolist()]                       # :372
        dst_out[r, :, 0] = [s_map.get(v, 0) for v in dst_ids[r].tolist()]                       # :378
        dst_out[r, :, 1] = d_cnt[d_inv]
    src_out[src_ids == 0] = 0.0
    dst_out[dst_ids == 0] = 0.0
    return src_out, dst_out


# ======================================================================================
# Dense path (PyTorch CPU fp32 functional ops; same operators the reference reaches)
# ======================================================================================

def _t(params: Dict[str, np.ndarray], key: str) -> torch.Tensor:
    v = params[key]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))


def time_encode(params, dt: torch.Tensor) -> torch.Tensor:
    """models/modules.py:27-39: cos(Linear(1,F_t)(dt)).

    The K=1 Linear is one multiply-add per element, and dt reaches ~2.7e6 s where one float32 ulp is
    0.25 rad: whether the host BLAS rounds the product before adding the bias changes cos() by up to
    0.25.  In the build container PyTorch's CPU addmm fuses it (bit-identical to fma(dt, w, b) on 320k
    samples, and what the golden vectors hold); other hosts' BLAS kernels do not.  The oracle therefore
    states the fused form explicitly so it gives the golden values on every host: the float64
    product of two float32 is exact, so float32(float64(dt)*w + b) is the single-rounding fma
    (up to a double rounding of the sum that cannot exceed one float32 ulp in ~1e-7 of the cases)."""
    w = _t(params, "time_encoder.w.weight").reshape(-1).double()
    b = _t(params, "time_encoder.w.bias").double()
    pre = (dt.double().unsqueeze(-1) * w + b).float()
    return torch.cos(pre)


def cooccurrence_features(params, counts: torch.Tensor) -> torch.Tensor:
    """models/DyGFormer.py:332-335 and :409-411: f(c0)+f(c1), f = Linear(1,C)->ReLU->Linear(C,C)."""
    pre = "neighbor_co_occurrence_encoder.neighbor_co_occurrence_encode_layer."
    h = F.relu(F.linear(counts.unsqueeze(-1), _t(params, pre + "0.weight"), _t(params, pre + "0.bias")))
    h = F.linear(h, _t(params, pre + "2.weight"), _t(params, pre + "2.bias"))
    return h.sum(dim=2)


def encoder_layer(params, layer: int, x: torch.Tensor, num_heads: int) -> torch.Tensor:
    """models/DyGFormer.py:442-461 in eval mode (dropout = identity).  Pre-LN; the attention is
    nn.MultiheadAttention's explicit path (need_weights=True default): packed in-proj, q scaled by
    1/sqrt(head_dim) before q.k^T, full softmax with NO mask, out-proj; exact-erf GELU FFN."""
    p = f"transformers.{layer}."
    B, T, D = x.shape
    hd = D // num_heads
    h = F.layer_norm(x, (D,), _t(params, p + "norm_layers.0.weight"), _t(params, p + "norm_layers.0.bias"), 1e-5)
    qkv = F.linear(h, _t(params, p + "multi_head_attention.in_proj_weight"), _t(params, p + "multi_head_attention.in_proj_bias"))
    q, k, v = qkv.split(D, dim=-1)
    q = q.reshape(B, T, num_heads, hd).transpose(1, 2) * math.sqrt(1.0 / float(hd))
    k = k.reshape(B, T, num_heads, hd).transpose(1, 2)
    v = v.reshape(B, T, num_heads, hd).transpose(1, 2)
    att = torch.softmax(q @ k.transpose(-2, -1), dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, T, D)
    o = F.linear(o, _t(params, p + "multi_head_attention.out_proj.weight"), _t(params, p + "multi_head_attention.out_proj.bias"))
    x = x + o                                                                                   # :456
    h = F.layer_norm(x, (D,), _t(params, p + "norm_layers.1.weight"), _t(params, p + "norm_layers.1.bias"), 1e-5)
    h = F.gelu(F.linear(h, _t(params, p + "linear_layers.0.weight"), _t(params, p + "linear_layers.0.bias")))
    h = F.linear(h, _t(params, p + "linear_layers.1.weight"), _t(params, p + "linear_layers.1.bias"))
    return x + h                                                                                # :460


def dygformer_forward(params: Dict[str, np.ndarray], node_feat: np.ndarray, edge_feat: np.ndarray,
                      adj: OracleAdjacency, src_ids: np.ndarray, dst_ids: np.ndarray, times: np.ndarray,
                      patch_size: int, max_input_sequence_length: int, num_heads: int = 2,
                      num_layers: int = 2, taps: Optional[dict] = None):
    """models/DyGFormer.py:68-194.  Returns (src_emb[B,F_n], dst_emb[B,F_n]) as float32 tensors.
    `taps`, when given, receives intermediate results for stage-level parity tests."""
    P, L = patch_size, max_input_sequence_length
    node_t = node_feat if isinstance(node_feat, torch.Tensor) else torch.from_numpy(node_feat)
    edge_t = edge_feat if isinstance(edge_feat, torch.Tensor) else torch.from_numpy(edge_feat)
    times = np.asarray(times, dtype=np.float64)

    sides = []
    for ids in (src_ids, dst_ids):                                                              # :78-100
        sides.append(first_hop_windows(adj, ids, times, P, L))
    (s_id, s_e, s_t), (d_id, d_e, d_t) = sides
    c_s, c_d = count_nodes_appearances(s_id, d_id)                                              # :104-106
    cooc = [cooccurrence_features(params, torch.from_numpy(c_s)), cooccurrence_features(params, torch.from_numpy(c_d))]

    chans = []
    for (pid, pe, pt), co in zip(sides, cooc):
        B, S = pid.shape
        nf = node_t[torch.from_numpy(pid)]                                                      # :259
        ef = edge_t[torch.from_numpy(pe)]                                                       # :261
        dt = torch.from_numpy(times[:, None] - pt).float()                                      # :263 (f64 - f32 -> f64 -> f32)
        tf = time_encode(params, dt)
        tf[torch.from_numpy(pid == 0)] = 0.0                                                    # :266
        Tn = S // P
        feats = [nf.reshape(B, Tn, -1), ef.reshape(B, Tn, -1), tf.reshape(B, Tn, -1), co.reshape(B, Tn, -1)]  # :288-304
        proj = []
        for name, f in zip(("node", "edge", "time", "neighbor_co_occurrence"), feats):          # :148-157
            proj.append(F.linear(f, _t(params, f"projection_layer.{name}.weight"), _t(params, f"projection_layer.{name}.bias")))
        chans.append(proj)
    Ts, Td = chans[0][0].shape[1], chans[1][0].shape[1]
    x = torch.stack([torch.cat([chans[0][c], chans[1][c]], dim=1) for c in range(4)], dim=2)     # :164-172
    B = x.shape[0]
    x = x.reshape(B, Ts + Td, -1)                                                               # :174
    if taps is not None:
        taps.update(src_ids=s_id, src_eids=s_e, src_times=s_t, dst_ids=d_id, dst_eids=d_e, dst_times=d_t,
                    src_counts=c_s, dst_counts=c_d, encoder_input=x.clone(), layer_outputs=[])
    for l in range(num_layers):                                                                 # :177-178
        x = encoder_layer(params, l, x, num_heads)
        if taps is not None:
            taps["layer_outputs"].append(x.clone())
    s = x[:, :Ts].mean(dim=1)                                                                   # :181-187
    d = x[:, Ts:Ts + Td].mean(dim=1)
    W, b = _t(params, "output_layer.weight"), _t(params, "output_layer.bias")
    return F.linear(s, W, b), F.linear(d, W, b)                                                 # :190-192


def merge_layer(mparams: Dict[str, np.ndarray], a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """models/modules.py:57-68: fc2(relu(fc1(cat(a,b))))."""
    x = torch.cat([a, b], dim=1)
    h = F.relu(F.linear(x, _t(mparams, "fc1.weight"), _t(mparams, "fc1.bias")))
    return F.linear(h, _t(mparams, "fc2.weight"), _t(mparams, "fc2.bias"))


def link_prediction_step(params, mparams, node_feat, edge_feat, adj, src, dst, neg_dst, times,
                         patch_size, max_input_sequence_length, num_heads=2, num_layers=2):
    """The forward body of evaluate_models_utils.py:126-141 for DyGFormer with `random` negatives
    (negative sources = batch sources, :62-63): two hot-path calls + MergeLayer + sigmoid on both."""
    with torch.no_grad():
        s, d = dygformer_forward(params, node_feat, edge_feat, adj, src, dst, times, patch_size,
                                 max_input_sequence_length, num_heads, num_layers)
        ns, nd = dygformer_forward(params, node_feat, edge_feat, adj, src, neg_dst, times, patch_size,
                                   max_input_sequence_length, num_heads, num_layers)
        pos = merge_layer(mparams, s, d).squeeze(-1).sigmoid()
        neg = merge_layer(mparams, ns, nd).squeeze(-1).sigmoid()
    return pos, neg
