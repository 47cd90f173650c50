"""Drop-in for the reference `TGAT` backbone (models/TGAT.py:9-147): same constructor, same
`compute_src_dst_node_temporal_embeddings(src_node_ids, dst_node_ids, node_interact_times, num_neighbors)` /
`compute_node_temporal_embeddings` / `set_neighbor_sampler` signatures, same parameter names (state_dict
compatible); the forward runs in libdygnn_hip.so (`dygnn_tgat_forward`).  Inference only, `recent` sampling."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _capi
from .modules import MergeLayer, TimeEncoder
from .neighbor_sampler import NeighborSampler


class MultiHeadAttention(nn.Module):
    """Parameters of models/modules.py:99-135 (bias-free q/k/v projections, LayerNorm, residual_fc)."""

    def __init__(self, node_feat_dim: int, edge_feat_dim: int, time_feat_dim: int, num_heads: int = 2, dropout: float = 0.1):
        super().__init__()
        self.node_feat_dim, self.edge_feat_dim, self.time_feat_dim, self.num_heads = node_feat_dim, edge_feat_dim, time_feat_dim, num_heads
        self.query_dim = node_feat_dim + time_feat_dim
        self.key_dim = node_feat_dim + edge_feat_dim + time_feat_dim
        assert self.query_dim % num_heads == 0, "The sum of node_feat_dim and time_feat_dim should be divided by num_heads!"
        self.head_dim = self.query_dim // num_heads
        self.query_projection = nn.Linear(self.query_dim, num_heads * self.head_dim, bias=False)
        self.key_projection = nn.Linear(self.key_dim, num_heads * self.head_dim, bias=False)
        self.value_projection = nn.Linear(self.key_dim, num_heads * self.head_dim, bias=False)
        self.scaling_factor = self.head_dim ** -0.5
        self.layer_norm = nn.LayerNorm(self.query_dim)
        self.residual_fc = nn.Linear(num_heads * self.head_dim, self.query_dim)
        self.dropout = nn.Dropout(dropout)


class TGAT(nn.Module):

    def __init__(self, node_raw_features: np.ndarray, edge_raw_features: np.ndarray, neighbor_sampler: NeighborSampler,
                 time_feat_dim: int, num_layers: int = 2, num_heads: int = 2, dropout: float = 0.1, device: str = "cpu"):
        super().__init__()
        self.node_raw_features = torch.from_numpy(np.ascontiguousarray(node_raw_features, dtype=np.float32)).to(device)
        self.edge_raw_features = torch.from_numpy(np.ascontiguousarray(edge_raw_features, dtype=np.float32)).to(device)
        self.neighbor_sampler = neighbor_sampler
        self.node_feat_dim = self.node_raw_features.shape[1]
        self.edge_feat_dim = self.edge_raw_features.shape[1]
        self.time_feat_dim = time_feat_dim
        self.num_layers = num_layers
        self.num_heads = num_heads
        self.dropout = dropout
        self.time_encoder = TimeEncoder(time_dim=time_feat_dim)
        self.temporal_conv_layers = nn.ModuleList([MultiHeadAttention(self.node_feat_dim, self.edge_feat_dim, self.time_feat_dim,
                                                                      self.num_heads, self.dropout) for _ in range(num_layers)])
        self.merge_layers = nn.ModuleList([MergeLayer(input_dim1=self.node_feat_dim + self.time_feat_dim, input_dim2=self.node_feat_dim,
                                                      hidden_dim=self.node_feat_dim, output_dim=self.node_feat_dim) for _ in range(num_layers)])
        self._lib = _capi.load()
        self._workspace: Dict[tuple, torch.Tensor] = {}

    def set_neighbor_sampler(self, neighbor_sampler: NeighborSampler):
        """models/TGAT.py:138-147."""
        self.neighbor_sampler = neighbor_sampler
        if self.neighbor_sampler.sample_neighbor_strategy in ["uniform", "time_interval_aware"]:
            assert self.neighbor_sampler.seed is not None
            self.neighbor_sampler.reset_random_state()

    def compute_src_dst_node_temporal_embeddings(self, src_node_ids, dst_node_ids, node_interact_times,
                                                 num_neighbors: int = 20) -> Tuple[torch.Tensor, torch.Tensor]:
        """models/TGAT.py:48-64: two float32 tensors [B, node_feat_dim] on the model's device."""
        if torch.is_grad_enabled() and (self.training or any(p.requires_grad for p in self.parameters())):
            # eval mode with autograd recording would return tensors without a graph: loss.backward() would silently do nothing
            raise NotImplementedError("TGAT forward with autograd recording (training) is not built on the HIP path (SURVEY.md §8f-1): "
                                      "call it under torch.no_grad()")
        random_strategy = self.neighbor_sampler.sample_neighbor_strategy != "recent"
        self.neighbor_sampler._check_strategy()
        dev = self.merge_layers[0].fc1.weight.device
        if dev.type != "cuda":
            raise _capi.DygnnError("dyglib_amd.TGAT runs on an MI355X only; there is no CPU fallback")
        if self.node_raw_features.device != dev:
            self.node_raw_features = self.node_raw_features.to(dev)
            self.edge_raw_features = self.edge_raw_features.to(dev)
        to_dev = lambda x, dt: (x.to(device=dev, dtype=dt).contiguous() if isinstance(x, torch.Tensor)
                                else torch.from_numpy(np.ascontiguousarray(x, dtype={torch.int64: np.int64, torch.float64: np.float64}[dt])).to(dev))
        csr = self.neighbor_sampler.csr
        if getattr(self, "_validated_csr", None) is not csr:          # once per sampler: every id reachable through the graph is inside the tables
            csr.check_tables(self.node_raw_features.shape[0], self.edge_raw_features.shape[0])
            self._validated_csr = csr
        csr.check_query_ids(src_node_ids, limit=self.node_raw_features.shape[0])       # IndexError like the reference (models/TGAT.py:85)
        csr.check_query_ids(dst_node_ids, limit=self.node_raw_features.shape[0])
        src, dst, tms = to_dev(src_node_ids, torch.int64), to_dev(dst_node_ids, torch.int64), to_dev(node_interact_times, torch.float64)
        B = src.numel()
        assert dst.numel() == B and tms.numel() == B
        out = torch.empty((2, B, self.node_feat_dim), dtype=torch.float32, device=dev)      # one block: the library writes it in place
        out_src, out_dst = out[0], out[1]
        if B == 0:
            return out_src, out_dst
        cfg, w = self._config_and_weights(num_neighbors)
        nbytes = self._lib.dygnn_tgat_workspace_bytes(C.byref(cfg), B)
        if nbytes == 0:
            _capi.check(-1)                      # AssertionError with the library's message (e.g. num_neighbors <= 0)
        key = (B, int(num_neighbors), torch.cuda.current_stream(dev).cuda_stream)
        ws = self._workspace.get(key)
        if ws is None or ws.numel() < nbytes or ws.device != dev:
            if len(self._workspace) > 8:
                self._workspace.clear()
            ws = self._workspace[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        if random_strategy:
            lv, keep = self._sample_levels_host(src.cpu().numpy(), dst.cpu().numpy(), tms.cpu().numpy(), int(num_neighbors), dev)
            _capi.check(self._lib.dygnn_tgat_forward_levels(C.byref(cfg), C.byref(w), C.byref(lv), self.node_raw_features.data_ptr(),
                                                            self.edge_raw_features.data_ptr(), B, out_src.data_ptr(), out_dst.data_ptr(),
                                                            ws.data_ptr(), ws.numel(), _capi.current_stream_ptr()))
            torch.cuda.current_stream(dev).synchronize()          # `keep` (the level tensors) may be freed afterwards
            return out_src, out_dst
        self._last_call = (cfg, B, ws)
        _capi.check(self._lib.dygnn_tgat_forward(C.byref(cfg), C.byref(w), self.neighbor_sampler.csr.on_device(dev),
                                                 self.node_raw_features.data_ptr(), self.edge_raw_features.data_ptr(),
                                                 src.data_ptr(), dst.data_ptr(), tms.data_ptr(), B, out_src.data_ptr(), out_dst.data_ptr(),
                                                 ws.data_ptr(), ws.numel(), _capi.current_stream_ptr()))
        return out_src, out_dst

    def compute_step_embeddings(self, src_node_ids, dst_node_ids, neg_dst_node_ids, node_interact_times, num_neighbors: int = 20):
        """The positive and the negative call of an evaluation step (evaluate_models_utils.py:126-136) as ONE library call on the roots
        [sources ; destinations ; negative destinations] at the batch times (dygnn_tgat_forward_roots): the negative call's sources are the
        positive call's (:62-63) and a root's row does not depend on the batch it is in, so (src_emb, dst_emb, neg_dst_emb) are bit-identical to
        compute_src_dst_node_temporal_embeddings(src, dst, t) and (src, neg_dst, t)[1].  `recent` sampling only (the random strategies consume the
        sampler's RandomState call by call)."""
        if self.neighbor_sampler.sample_neighbor_strategy != "recent":
            raise NotImplementedError("compute_step_embeddings: `recent` sampling only; issue the two calls of the reference for the random strategies")
        if torch.is_grad_enabled() and (self.training or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("TGAT forward with autograd recording (training) is not built on the HIP path (SURVEY.md §8f-1): "
                                      "call it under torch.no_grad()")
        dev = self.merge_layers[0].fc1.weight.device
        if dev.type != "cuda":
            raise _capi.DygnnError("dyglib_amd.TGAT runs on an MI355X only; there is no CPU fallback")
        if self.node_raw_features.device != dev:
            self.node_raw_features = self.node_raw_features.to(dev)
            self.edge_raw_features = self.edge_raw_features.to(dev)
        csr = self.neighbor_sampler.csr
        if getattr(self, "_validated_csr", None) is not csr:
            csr.check_tables(self.node_raw_features.shape[0], self.edge_raw_features.shape[0])
            self._validated_csr = csr
        parts_i, parts_t = [], []
        for ids in (src_node_ids, dst_node_ids, neg_dst_node_ids):
            csr.check_query_ids(ids, limit=self.node_raw_features.shape[0])
            parts_i.append(ids.to(device=dev, dtype=torch.int64) if isinstance(ids, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int64)).to(dev))
        tms = (node_interact_times.to(device=dev, dtype=torch.float64) if isinstance(node_interact_times, torch.Tensor)
               else torch.from_numpy(np.ascontiguousarray(node_interact_times, dtype=np.float64)).to(dev))
        B = parts_i[0].numel()
        assert parts_i[1].numel() == B and parts_i[2].numel() == B and tms.numel() == B
        pad = (3 * B) % 2                                   # the library takes an even number of roots: repeat the last one
        roots = torch.cat(parts_i + ([parts_i[2][-1:]] if pad and B else []))
        times = torch.cat([tms, tms, tms] + ([tms[-1:]] if pad and B else []))
        n = roots.numel()
        out = torch.empty((n, self.node_feat_dim), dtype=torch.float32, device=dev)
        if B == 0:
            return out[:0], out[:0], out[:0]
        cfg, w = self._config_and_weights(num_neighbors)
        nbytes = self._lib.dygnn_tgat_workspace_bytes(C.byref(cfg), n // 2)
        if nbytes == 0:
            _capi.check(-1)
        key = (n // 2, int(num_neighbors), torch.cuda.current_stream(dev).cuda_stream)
        ws = self._workspace.get(key)
        if ws is None or ws.numel() < nbytes or ws.device != dev:
            if len(self._workspace) > 8:
                self._workspace.clear()
            ws = self._workspace[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self._last_call = (cfg, n // 2, ws)
        _capi.check(self._lib.dygnn_tgat_forward_roots(C.byref(cfg), C.byref(w), csr.on_device(dev), self.node_raw_features.data_ptr(),
                                                       self.edge_raw_features.data_ptr(), roots.data_ptr(), times.data_ptr(), n, out.data_ptr(),
                                                       ws.data_ptr(), ws.numel(), _capi.current_stream_ptr()))
        return out[:B], out[B:2 * B], out[2 * B:3 * B]

    def _config_and_weights(self, num_neighbors: int):
        cfg = _capi.TgatConfig(self.node_feat_dim, self.edge_feat_dim, self.time_feat_dim, self.num_layers, self.num_heads, int(num_neighbors))
        w = _capi.TgatWeights()
        w.time_w, w.time_b = self.time_encoder.w.weight.data_ptr(), self.time_encoder.w.bias.data_ptr()
        for l in range(self.num_layers):
            a, m, L = self.temporal_conv_layers[l], self.merge_layers[l], w.layers[l]
            L.query_w, L.key_w, L.value_w = a.query_projection.weight.data_ptr(), a.key_projection.weight.data_ptr(), a.value_projection.weight.data_ptr()
            L.ln_w, L.ln_b = a.layer_norm.weight.data_ptr(), a.layer_norm.bias.data_ptr()
            L.res_w, L.res_b = a.residual_fc.weight.data_ptr(), a.residual_fc.bias.data_ptr()
            L.fc1_w, L.fc1_b, L.fc2_w, L.fc2_b = m.fc1.weight.data_ptr(), m.fc1.bias.data_ptr(), m.fc2.weight.data_ptr(), m.fc2.bias.data_ptr()
        return cfg, w

    def compute_node_temporal_embeddings(self, node_ids, node_interact_times, current_layer_num: int, num_neighbors: int = 20) -> torch.Tensor:
        """models/TGAT.py:66-136: the embedding of `node_ids` at `node_interact_times` after `current_layer_num` layers ([n, node_feat_dim]).
        Layer 0 is the raw node feature row (:85-88); layer l is the l-layer model over the first l conv / merge layers (the recursion only
        ever descends, :92-110), i.e. one library call with num_layers = l on the nodes as both sides (`recent`: duplicates are computed once)."""
        assert current_layer_num >= 0                                                      # models/TGAT.py:77
        if current_layer_num > self.num_layers:
            raise IndexError("index out of range in temporal_conv_layers")              # ModuleList indexing in the reference (:123)
        if current_layer_num == 0:
            self.neighbor_sampler.csr.check_query_ids(node_ids, limit=self.node_raw_features.shape[0])
            idx = node_ids if isinstance(node_ids, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(node_ids, dtype=np.int64))
            return self.node_raw_features[idx.to(self.node_raw_features.device)]
        full = self.num_layers
        try:
            self.num_layers = int(current_layer_num)
            emb, _ = self.compute_src_dst_node_temporal_embeddings(node_ids, node_ids, node_interact_times, num_neighbors=num_neighbors)
        finally:
            self.num_layers = full
        return emb

    def last_level_entries(self) -> Tuple[int, int]:
        """(entries over the computed levels, entries actually computed) of the last `recent` call: two-layer models compute every
        distinct (node, time) entry of level 1 once (dygnn_tgat_level_entries; synchronises the stream)."""
        cfg, B, ws = self._last_call
        total, computed = C.c_int64(0), C.c_int64(0)
        _capi.check(self._lib.dygnn_tgat_level_entries(C.byref(cfg), B, ws.data_ptr(), C.byref(total), C.byref(computed), _capi.current_stream_ptr()))
        return int(total.value), int(computed.value)

    # ---- random sampling strategies: the draws are replayed on the host in the reference's recursion order ----------------
    def _sample_levels_host(self, src: np.ndarray, dst: np.ndarray, t: np.ndarray, k: int, dev):
        """Level sets for dygnn_tgat_forward_levels.  models/TGAT.py:92-110: compute_node_temporal_embeddings(nodes, l) first
        recurses for the nodes themselves at layer l-1 (drawing THEIR neighbours), then draws the layer-l neighbours, then
        recurses for those; src is processed completely before dst (models/TGAT.py:57-62).  With a random sampler every one of
        those draws is independent and must consume the RandomState in exactly that order."""
        if self.num_layers not in (1, 2):
            raise NotImplementedError("TGAT with a random sampling strategy is built for num_layers 1 and 2")
        smp = self.neighbor_sampler
        L = self.num_layers

        def draw(nodes, times):
            n, e, tn = smp.get_historical_neighbors(nodes, times, num_neighbors=k)            # host round trip, RandomState replay
            dt = (times[:, None] - tn).astype(np.float32)                                     # models/TGAT.py:116-119
            return n, e, tn, dt

        per_side = []
        for nodes in (src, dst):
            if L == 1:
                per_side.append({"top": draw(nodes, t)})
            else:
                d1 = draw(nodes, t)                                        # neighbours of the nodes themselves, for their layer-1 embedding
                d2 = draw(nodes, t)                                        # layer-2 neighbours
                d3 = draw(d2[0].reshape(-1), d2[2].reshape(-1).astype(np.float64))           # neighbours of those, for THEIR layer-1 embedding
                per_side.append({"self": d1, "top": d2, "nbr": d3})
        s_, d_ = per_side
        ids = {L: np.concatenate([src, dst])}
        eid, dts = {}, {}
        eid[L] = np.concatenate([s_["top"][1], d_["top"][1]])
        dts[L] = np.concatenate([s_["top"][3], d_["top"][3]])
        ids[L - 1] = np.concatenate([ids[L], s_["top"][0].reshape(-1), d_["top"][0].reshape(-1)])
        if L == 2:
            eid[1] = np.concatenate([s_["self"][1], d_["self"][1], s_["nbr"][1], d_["nbr"][1]])
            dts[1] = np.concatenate([s_["self"][3], d_["self"][3], s_["nbr"][3], d_["nbr"][3]])
            nb1 = np.concatenate([s_["self"][0], d_["self"][0], s_["nbr"][0], d_["nbr"][0]])
            ids[0] = np.concatenate([ids[1], nb1.reshape(-1)])
        lv = _capi.TgatLevels()
        keep = []
        for l in range(L + 1):
            a = torch.from_numpy(np.ascontiguousarray(ids[l], dtype=np.int32)).to(dev)
            keep.append(a)
            lv.ids[l] = a.data_ptr()
            if l >= 1:
                b = torch.from_numpy(np.ascontiguousarray(eid[l], dtype=np.int32)).to(dev)
                c = torch.from_numpy(np.ascontiguousarray(dts[l], dtype=np.float32)).to(dev)
                keep += [b, c]
                lv.nbr_eid[l], lv.nbr_dt[l] = b.data_ptr(), c.data_ptr()
        return lv, keep
