"""Drop-in for the reference `MemoryModel` with model_name == 'TGN' (models/MemoryModel.py:9-168): same constructor,
`compute_src_dst_node_temporal_embeddings(src_node_ids, dst_node_ids, node_interact_times, edge_ids,
edges_are_positive, num_neighbors)`, `set_neighbor_sampler`, `memory_bank.__init_memory_bank__ / backup_memory_bank /
reload_memory_bank`, and the same state_dict keys.  The forward, the GRU memory update and the raw-message bookkeeping run
in libdygnn_hip.so (`dygnn_tgn_forward`).  JODIE / DyRep are not built (not in BASELINE.json's configs)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _capi
from .modules import MergeLayer, TimeEncoder
from .neighbor_sampler import NeighborSampler
from .tgat import MultiHeadAttention


class MemoryBank(nn.Module):
    """memory_bank.{node_memories,node_last_updated_times} are (non-trainable) Parameters like the reference's
    (MemoryModel.py:317-320).  The per-node Python list of raw messages becomes three device buffers holding the LAST
    pending message of every node: the aggregator only reads the last element (:284-291) and lists are cleared whole."""

    def __init__(self, num_nodes: int, memory_dim: int, message_dim: int):
        super().__init__()
        self.num_nodes, self.memory_dim, self.message_dim = num_nodes, memory_dim, message_dim
        self.node_memories = nn.Parameter(torch.zeros((num_nodes, memory_dim)), requires_grad=False)
        self.node_last_updated_times = nn.Parameter(torch.zeros(num_nodes), requires_grad=False)
        self.msg = self.msg_time = self.has_msg = None
        self.__init_memory_bank__()

    def _alloc(self):
        dev = self.node_memories.device
        if self.msg is None or self.msg.device != dev:
            self.msg = torch.zeros((self.num_nodes, self.message_dim), dtype=torch.float32, device=dev)
            self.msg_time = torch.zeros(self.num_nodes, dtype=torch.float64, device=dev)
            self.has_msg = torch.zeros(self.num_nodes, dtype=torch.int32, device=dev)

    def __init_memory_bank__(self):
        """MemoryModel.py:322-329: called at the start of every epoch."""
        self.node_memories.data.zero_()
        self.node_last_updated_times.data.zero_()
        self.msg = None
        self._alloc()

    def backup_memory_bank(self):
        """MemoryModel.py:345-354."""
        self._alloc()
        return (self.node_memories.data.clone(), self.node_last_updated_times.data.clone(),
                (self.msg.clone(), self.msg_time.clone(), self.has_msg.clone()))

    def reload_memory_bank(self, backup_memory_bank: tuple):
        """MemoryModel.py:356-366."""
        self.node_memories.data, self.node_last_updated_times.data = backup_memory_bank[0].clone(), backup_memory_bank[1].clone()
        self.msg, self.msg_time, self.has_msg = (x.clone() for x in backup_memory_bank[2])

    def detach_memory_bank(self):
        """MemoryModel.py:368-378: nothing to detach, the HIP path builds no autograd graph."""

    def get_memories(self, node_ids: np.ndarray):
        return self.node_memories[torch.from_numpy(np.asarray(node_ids))]


class GRUMemoryUpdater(nn.Module):
    """memory_updater.memory_updater = nn.GRUCell(message_dim, memory_dim) (MemoryModel.py:490-501); the memory bank is
    registered under it too, as in the reference, so the state_dict keys coincide."""

    def __init__(self, memory_bank: MemoryBank, message_dim: int, memory_dim: int):
        super().__init__()
        self.memory_bank = memory_bank
        self.memory_updater = nn.GRUCell(input_size=message_dim, hidden_size=memory_dim)


class GraphAttentionEmbedding(nn.Module):
    """Parameters of MemoryModel.py:548-578."""

    def __init__(self, node_feat_dim, edge_feat_dim, time_feat_dim, num_layers, num_heads, dropout, neighbor_sampler, time_encoder):
        super().__init__()
        self.neighbor_sampler = neighbor_sampler
        self.time_encoder = time_encoder              # the SAME module as the model's (MemoryModel.py:567): shared state_dict entries
        self.temporal_conv_layers = nn.ModuleList([MultiHeadAttention(node_feat_dim, edge_feat_dim, time_feat_dim, num_heads, dropout)
                                                   for _ in range(num_layers)])
        self.merge_layers = nn.ModuleList([MergeLayer(input_dim1=node_feat_dim + time_feat_dim, input_dim2=node_feat_dim,
                                                      hidden_dim=node_feat_dim, output_dim=node_feat_dim) for _ in range(num_layers)])


class MemoryModel(nn.Module):

    def __init__(self, node_raw_features: np.ndarray, edge_raw_features: np.ndarray, neighbor_sampler: NeighborSampler,
                 time_feat_dim: int, model_name: str = "TGN", num_layers: int = 2, num_heads: int = 2, dropout: float = 0.1,
                 src_node_mean_time_shift: float = 0.0, src_node_std_time_shift: float = 1.0, dst_node_mean_time_shift_dst: float = 0.0,
                 dst_node_std_time_shift: float = 1.0, device: str = "cpu"):
        super().__init__()
        if model_name in ("DyRep", "JODIE"):
            raise NotImplementedError(f"model_name {model_name} is not built on the HIP path (BASELINE config 5 is TGN)")
        if model_name != "TGN":
            raise ValueError(f"Not implemented error for model_name {model_name}!")           # MemoryModel.py:63
        self.node_raw_features = torch.from_numpy(np.ascontiguousarray(node_raw_features, dtype=np.float32)).to(device)
        self.edge_raw_features = torch.from_numpy(np.ascontiguousarray(edge_raw_features, dtype=np.float32)).to(device)
        self.node_feat_dim, self.edge_feat_dim = self.node_raw_features.shape[1], self.edge_raw_features.shape[1]
        self.time_feat_dim, self.num_layers, self.num_heads, self.dropout, self.device = time_feat_dim, num_layers, num_heads, dropout, device
        self.model_name = model_name
        self.num_nodes = self.node_raw_features.shape[0]
        self.memory_dim = self.node_feat_dim
        self.message_dim = self.memory_dim + self.memory_dim + self.time_feat_dim + self.edge_feat_dim
        self.time_encoder = TimeEncoder(time_dim=time_feat_dim)
        self.memory_bank = MemoryBank(self.num_nodes, self.memory_dim, self.message_dim)
        self.memory_updater = GRUMemoryUpdater(self.memory_bank, self.message_dim, self.memory_dim)
        self.embedding_module = GraphAttentionEmbedding(self.node_feat_dim, self.edge_feat_dim, self.time_feat_dim, num_layers, num_heads,
                                                        dropout, neighbor_sampler, self.time_encoder)
        self._lib = _capi.load()
        self._workspace: Dict[tuple, torch.Tensor] = {}

    def set_neighbor_sampler(self, neighbor_sampler: NeighborSampler):
        """MemoryModel.py:253-263."""
        self.embedding_module.neighbor_sampler = neighbor_sampler
        if neighbor_sampler.sample_neighbor_strategy in ["uniform", "time_interval_aware"]:
            assert neighbor_sampler.seed is not None
            neighbor_sampler.reset_random_state()

    def compute_step_embeddings(self, src_node_ids, dst_node_ids, neg_src_node_ids, neg_dst_node_ids, node_interact_times, edge_ids,
                                num_neighbors: int = 20):
        """The negative call and the positive call of one batch (evaluate_models_utils.py:85-107) as ONE library call: both read the
        same state and only the positive call writes it, at its end, so [positives ; negatives] is one batch whose first half updates the
        memory bank (dygnn_tgn_forward_step).  Returns (src_emb, dst_emb, neg_src_emb, neg_dst_emb), bit-identical to
        compute_src_dst_node_temporal_embeddings(neg..., edges_are_positive=False) followed by (...pos..., edges_are_positive=True)."""
        cat = lambda a, b: (torch.cat([a, b]) if isinstance(a, torch.Tensor) else np.concatenate([np.asarray(a), np.asarray(b)]))
        n_pos = len(src_node_ids)
        if self.embedding_module.neighbor_sampler.sample_neighbor_strategy != "recent":
            # random strategies draw per call: the reference's two calls in its order, negative first (evaluate_models_utils.py:85-107)
            ns, nd = self.compute_src_dst_node_temporal_embeddings(neg_src_node_ids, neg_dst_node_ids, node_interact_times, None, edges_are_positive=False,
                                                                   num_neighbors=num_neighbors)
            ps, pd = self.compute_src_dst_node_temporal_embeddings(src_node_ids, dst_node_ids, node_interact_times, edge_ids, edges_are_positive=True,
                                                                   num_neighbors=num_neighbors)
            return ps, pd, ns, nd
        s, d = self.compute_src_dst_node_temporal_embeddings(cat(src_node_ids, neg_src_node_ids), cat(dst_node_ids, neg_dst_node_ids),
                                                             cat(node_interact_times, node_interact_times), edge_ids, edges_are_positive=True,
                                                             num_neighbors=num_neighbors, _n_positive=n_pos)
        return s[:n_pos], d[:n_pos], s[n_pos:], d[n_pos:]

    def compute_step_embeddings_joint(self, src_pos_neg, dst_pos_neg, times_pos_neg, edge_ids, n_positive: int, num_neighbors: int = 20):
        """compute_step_embeddings for a caller that already holds the step as ONE batch [positives ; negatives] (ids and times [2B], edge
        ids [B]): returns (src_emb, dst_emb) [2B, dim] as the library wrote them — rows 0 .. n_positive-1 are the positive call's, the rest the
        negative call's — so neither the inputs nor the link predictor's operands are concatenated on the device per step."""
        if self.embedding_module.neighbor_sampler.sample_neighbor_strategy != "recent":
            raise NotImplementedError("compute_step_embeddings_joint: `recent` sampling only (a random strategy draws per call: use compute_step_embeddings)")
        return self.compute_src_dst_node_temporal_embeddings(src_pos_neg, dst_pos_neg, times_pos_neg, edge_ids, edges_are_positive=True,
                                                             num_neighbors=num_neighbors, _n_positive=int(n_positive))

    def compute_src_dst_node_temporal_embeddings(self, src_node_ids, dst_node_ids, node_interact_times, edge_ids,
                                                 edges_are_positive: bool = True, num_neighbors: int = 20, _n_positive: int = None
                                                 ) -> Tuple[torch.Tensor, torch.Tensor]:
        """MemoryModel.py:87-168 (TGN).  Positive calls mutate the memory bank: issue batches in chronological order."""
        if torch.is_grad_enabled() and (self.training or any(p.requires_grad for p in self.parameters())):
            # eval mode with autograd recording would return tensors without a graph: loss.backward() would silently do nothing
            raise NotImplementedError("MemoryModel forward with autograd recording (training) is not built on the HIP path (SURVEY.md §8f-1): "
                                      "call it under torch.no_grad()")
        sampler = self.embedding_module.neighbor_sampler
        random_strategy = sampler.sample_neighbor_strategy != "recent"
        if random_strategy and _n_positive is not None:
            raise NotImplementedError("a joint [positives ; negatives] step needs `recent` sampling (a random strategy draws per call)")
        dev = self.memory_bank.node_memories.device
        if dev.type != "cuda":
            raise _capi.DygnnError("dyglib_amd.MemoryModel runs on an MI355X only; there is no CPU fallback")
        if self.node_raw_features.device != dev:
            self.node_raw_features, self.edge_raw_features = self.node_raw_features.to(dev), self.edge_raw_features.to(dev)
        self.memory_bank._alloc()
        to_dev = lambda x, dt: (x.to(device=dev, dtype=dt).contiguous() if isinstance(x, torch.Tensor)
                                else torch.from_numpy(np.ascontiguousarray(x, dtype={torch.int64: np.int64, torch.float64: np.float64}[dt])).to(dev))
        csr = sampler.csr
        if getattr(self, "_validated_csr", None) is not csr:          # once per sampler: graph ids inside the tables and the memory bank
            csr.check_tables(min(self.node_raw_features.shape[0], self.num_nodes), self.edge_raw_features.shape[0])
            self._validated_csr = csr
        csr.check_query_ids(src_node_ids, limit=self.num_nodes)        # IndexError like the reference's memory_bank indexing (MemoryModel.py:336)
        csr.check_query_ids(dst_node_ids, limit=self.num_nodes)
        if edge_ids is not None and not isinstance(edge_ids, torch.Tensor) and len(edge_ids):
            e = np.asarray(edge_ids)
            if int(e.min()) < 0 or int(e.max()) >= self.edge_raw_features.shape[0]:
                raise IndexError(f"edge id out of bounds for edge_raw_features with {self.edge_raw_features.shape[0]} rows")
        src, dst, tms = to_dev(src_node_ids, torch.int64), to_dev(dst_node_ids, torch.int64), to_dev(node_interact_times, torch.float64)
        B = src.numel()
        assert dst.numel() == B and tms.numel() == B
        if edges_are_positive:
            assert edge_ids is not None                                                        # MemoryModel.py:140
        eids = to_dev(edge_ids, torch.int64) if edge_ids is not None else None
        out = torch.empty((2, B, self.node_feat_dim), dtype=torch.float32, device=dev)      # one block: the library writes it in place
        out_src, out_dst = out[0], out[1]
        if B == 0:
            return out_src, out_dst
        cfg = _capi.TgatConfig(self.node_feat_dim, self.edge_feat_dim, self.time_feat_dim, self.num_layers, self.num_heads, int(num_neighbors))
        w = _capi.TgatWeights()
        w.time_w, w.time_b = self.time_encoder.w.weight.data_ptr(), self.time_encoder.w.bias.data_ptr()
        em = self.embedding_module
        for l in range(self.num_layers):
            a, m, L = em.temporal_conv_layers[l], em.merge_layers[l], w.layers[l]
            L.query_w, L.key_w, L.value_w = a.query_projection.weight.data_ptr(), a.key_projection.weight.data_ptr(), a.value_projection.weight.data_ptr()
            L.ln_w, L.ln_b = a.layer_norm.weight.data_ptr(), a.layer_norm.bias.data_ptr()
            L.res_w, L.res_b = a.residual_fc.weight.data_ptr(), a.residual_fc.bias.data_ptr()
            L.fc1_w, L.fc1_b, L.fc2_w, L.fc2_b = m.fc1.weight.data_ptr(), m.fc1.bias.data_ptr(), m.fc2.weight.data_ptr(), m.fc2.bias.data_ptr()
        cell = self.memory_updater.memory_updater
        gru = _capi.GruWeights(cell.weight_ih.data_ptr(), cell.weight_hh.data_ptr(), cell.bias_ih.data_ptr(), cell.bias_hh.data_ptr())
        mb = self.memory_bank
        st = _capi.TgnState(self.num_nodes, mb.node_memories.data_ptr(), mb.node_last_updated_times.data_ptr(), mb.msg.data_ptr(),
                            mb.msg_time.data_ptr(), mb.has_msg.data_ptr())
        nbytes = self._lib.dygnn_tgn_workspace_bytes(C.byref(cfg), self.num_nodes, B)
        if nbytes == 0:
            _capi.check(-1)
        key = (B, int(num_neighbors), torch.cuda.current_stream(dev).cuda_stream)
        ws = self._workspace.get(key)
        if ws is None or ws.numel() < nbytes or ws.device != dev:
            if len(self._workspace) > 8:
                self._workspace.clear()
            ws = self._workspace[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        n_pos = (B if edges_are_positive else 0) if _n_positive is None else int(_n_positive)
        if eids is not None and eids.numel() < n_pos:
            raise AssertionError("edge_ids must cover the positive edges")
        if random_strategy:
            # MemoryModel.py:626-629 with `uniform` / `time_interval_aware`: the draws are replayed on the host in the reference's order and the
            # library runs on the pre-sampled levels (dygnn_tgn_forward_levels)
            h = lambda x: x.cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
            lv, keep = self._sample_levels_host(h(src_node_ids).astype(np.int64), h(dst_node_ids).astype(np.int64), h(node_interact_times).astype(np.float64),
                                                int(num_neighbors), dev)
            _capi.check(self._lib.dygnn_tgn_forward_levels(C.byref(cfg), C.byref(w), C.byref(gru), C.byref(lv), self.node_raw_features.data_ptr(),
                                                           self.edge_raw_features.data_ptr(), C.byref(st), src.data_ptr(), dst.data_ptr(), tms.data_ptr(),
                                                           eids.data_ptr() if eids is not None else None, B, n_pos, out_src.data_ptr(), out_dst.data_ptr(),
                                                           ws.data_ptr(), ws.numel(), _capi.current_stream_ptr()))
            torch.cuda.current_stream(dev).synchronize()          # `keep` (the level tensors) may be freed afterwards
            return out_src, out_dst
        _capi.check(self._lib.dygnn_tgn_forward_step(C.byref(cfg), C.byref(w), C.byref(gru), sampler.csr.on_device(dev),
                                                     self.node_raw_features.data_ptr(), self.edge_raw_features.data_ptr(), C.byref(st),
                                                     src.data_ptr(), dst.data_ptr(), tms.data_ptr(), eids.data_ptr() if eids is not None else None,
                                                     B, n_pos, out_src.data_ptr(), out_dst.data_ptr(),
                                                     ws.data_ptr(), ws.numel(), _capi.current_stream_ptr()))
        return out_src, out_dst

    # ---- random sampling strategies: the draws are replayed on the host in the reference's recursion order -------------------------------
    def _sample_levels_host(self, src: np.ndarray, dst: np.ndarray, t: np.ndarray, k: int, dev):
        """Level sets for dygnn_tgn_forward_levels.  The reference embeds [src ; dst] in ONE recursion (MemoryModel.py:104-131):
        compute_node_temporal_embeddings(nodes, l) first recurses for the nodes themselves at layer l-1 (drawing THEIR neighbours, :596-600),
        then draws the layer-l neighbours (:606-609), then recurses for those (:613-617).  Every draw consumes the sampler's RandomState."""
        if self.num_layers not in (1, 2):
            raise NotImplementedError("TGN with a random sampling strategy is built for num_layers 1 and 2")
        smp = self.embedding_module.neighbor_sampler
        L = self.num_layers
        nodes, times = np.concatenate([src, dst]), np.concatenate([t, t])

        def draw(n_, t_):
            n, e, tn = smp.get_historical_neighbors(n_, t_, num_neighbors=k)               # host round trip, RandomState replay
            return n, e, tn, (t_[:, None] - tn).astype(np.float32)                           # MemoryModel.py:623-626
        ids, eid, dts = {L: nodes}, {}, {}
        if L == 1:
            top = draw(nodes, times)
        else:
            own = draw(nodes, times)                                   # neighbours of the nodes themselves, for their layer-1 embedding
            top = draw(nodes, times)                                   # layer-2 neighbours
            nbr = draw(top[0].reshape(-1), top[2].reshape(-1).astype(np.float64))           # neighbours of those, for THEIR layer-1 embedding
            eid[1], dts[1] = np.concatenate([own[1], nbr[1]]), np.concatenate([own[3], nbr[3]])
        eid[L], dts[L] = top[1], top[3]
        ids[L - 1] = np.concatenate([nodes, top[0].reshape(-1)])
        if L == 2:
            ids[0] = np.concatenate([ids[1], own[0].reshape(-1), nbr[0].reshape(-1)])
        lv, keep = _capi.TgatLevels(), []
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
