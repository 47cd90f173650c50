"""Drop-in for the reference `DyGFormer` backbone (models/DyGFormer.py:11-317): same constructor,
same `compute_src_dst_node_temporal_embeddings` / `set_neighbor_sampler` signatures, same
parameter names and shapes (so `state_dict` round-trips with reference checkpoints) — but the whole
forward (neighbour windows, co-occurrence counts, feature gathers, time encoding, patch
projection, both encoder layers, pooling, output layer) runs inside libdygnn_hip.so with no host
round trip: the reference's numpy hop (sampling/padding/counting on the host CPU,
models/DyGFormer.py:78-106) disappears and ids never leave the GPU.

Inference (eval mode, or no_grad) runs the fused kernel.  In training mode — or whenever autograd is
recording in eval mode — the call goes through `_TrainFunction`: the training forward of
dygformer_train.hip (dropout from a counter-based generator, activations kept in a per-call
workspace) and its hand-written backward pass, so `loss.backward()` / `optimizer.step()` of
train_link_prediction.py:229-257 work unchanged.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
from torch.nn import MultiheadAttention

from . import _capi
from .modules import TimeEncoder
from .neighbor_sampler import NeighborSampler


class NeighborCooccurrenceEncoder(nn.Module):
    """Parameters of models/DyGFormer.py:320-335 (Linear(1,C) -> ReLU -> Linear(C,C))."""

    def __init__(self, neighbor_co_occurrence_feat_dim: int, device: str = "cpu"):
        super().__init__()
        self.neighbor_co_occurrence_feat_dim = neighbor_co_occurrence_feat_dim
        self.device = device
        self.neighbor_co_occurrence_encode_layer = nn.Sequential(
            nn.Linear(in_features=1, out_features=neighbor_co_occurrence_feat_dim),
            nn.ReLU(),
            nn.Linear(in_features=neighbor_co_occurrence_feat_dim, out_features=neighbor_co_occurrence_feat_dim))


class TransformerEncoder(nn.Module):
    """Parameters of models/DyGFormer.py:418-440 (pre-LN encoder layer)."""

    def __init__(self, attention_dim: int, num_heads: int, dropout: float = 0.1):
        super().__init__()
        self.multi_head_attention = MultiheadAttention(embed_dim=attention_dim, num_heads=num_heads, dropout=dropout)
        self.dropout = nn.Dropout(dropout)
        self.linear_layers = nn.ModuleList([nn.Linear(attention_dim, 4 * attention_dim), nn.Linear(4 * attention_dim, attention_dim)])
        self.norm_layers = nn.ModuleList([nn.LayerNorm(attention_dim), nn.LayerNorm(attention_dim)])


class _TrainFunction(torch.autograd.Function):
    """compute_src_dst_node_temporal_embeddings with gradients: forward = dygnn_dygformer_train_forward, backward =
    dygnn_dygformer_backward.  The parameters are passed as inputs only so that autograd routes their gradients."""

    @staticmethod
    def forward(ctx, model, src, dst, tms, dropout_p, seed, seq_lens, *params):
        dev = src.device
        B = src.numel()
        lib, cfg = model._lib, model._cfg
        # the fused training forward reads the kernel-ready weight copy (refreshed here after an optimizer step, launches only)
        weights, packed = model._packed_weights(dev, training=True)
        nbytes = lib.dygnn_dygformer_train_workspace_bytes(C.byref(cfg), B)
        if nbytes == 0:
            _capi.check(-3)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)          # lives until this call's backward
        out_src = torch.empty((B, model.node_feat_dim), dtype=torch.float32, device=dev)
        out_dst = torch.empty_like(out_src)
        seq = (C.c_int32 * 2)()
        if seq_lens is not None:
            seq[0], seq[1] = int(seq_lens[0]), int(seq_lens[1])
        csr = model.neighbor_sampler.csr.on_device(dev)
        rc = lib.dygnn_dygformer_train_forward(C.byref(cfg), C.byref(weights), csr, model.node_raw_features.data_ptr(),
                                               model.edge_raw_features.data_ptr(), src.data_ptr(), dst.data_ptr(), tms.data_ptr(), B,
                                               float(dropout_p), int(seed), out_src.data_ptr(), out_dst.data_ptr(), ws.data_ptr(), nbytes,
                                               C.cast(seq, C.c_void_p), packed.data_ptr(), _capi.current_stream_ptr())
        _capi.check(rc)
        ctx.model, ctx.ws, ctx.seq, ctx.B, ctx.dropout_p, ctx.seed = model, ws, seq, B, float(dropout_p), int(seed)
        ctx.packed, ctx.weights = packed, weights
        ctx.param_versions = [(p.data_ptr(), p._version) for p in model._param_list()]
        return out_src, out_dst

    @staticmethod
    def backward(ctx, g_src, g_dst):
        model = ctx.model
        dev = ctx.ws.device
        g_src = (g_src if g_src is not None else torch.zeros((ctx.B, model.node_feat_dim), device=dev)).contiguous().float()
        g_dst = (g_dst if g_dst is not None else torch.zeros((ctx.B, model.node_feat_dim), device=dev)).contiguous().float()
        params = model._param_list()
        # the backward pass re-reads the CURRENT parameter values: they must be the ones the forward used (PyTorch raises the same
        # way when a tensor saved for backward was modified in place, e.g. by an optimizer step between forward and backward)
        if [(p.data_ptr(), p._version) for p in params] != ctx.param_versions:
            raise RuntimeError("one of the variables needed for gradient computation has been modified by an inplace operation: "
                               "a DyGFormer parameter changed between this call's forward and its backward")
        sizes = model.__dict__.get("_psizes")
        if sizes is None:
            sizes = model.__dict__["_psizes"] = [p.numel() for p in params]
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)       # one fill for all gradient buffers
        grads = [g.view_as(p) for g, p in zip(flat.split(sizes), params)]
        gstruct = model._weights_struct(grads)
        weights = ctx.weights
        rc = model._lib.dygnn_dygformer_backward(C.byref(model._cfg), C.byref(weights), C.byref(gstruct), g_src.data_ptr(), g_dst.data_ptr(), ctx.B,
                                                 ctx.dropout_p, ctx.seed, C.cast(ctx.seq, C.c_void_p), ctx.ws.data_ptr(), ctx.ws.numel(),
                                                 ctx.packed.data_ptr(), _capi.current_stream_ptr())
        _capi.check(rc)
        ctx.ws = None
        return (None, None, None, None, None, None, None, *grads)


class DyGFormer(nn.Module):

    def __init__(self, node_raw_features: np.ndarray, edge_raw_features: np.ndarray, neighbor_sampler: NeighborSampler,
                 time_feat_dim: int, channel_embedding_dim: int, patch_size: int = 1, num_layers: int = 2, num_heads: int = 2,
                 dropout: float = 0.1, max_input_sequence_length: int = 512, device: str = "cpu"):
        super().__init__()
        # plain attributes, not buffers: they are not part of the state_dict (models/DyGFormer.py:32-33)
        self.node_raw_features = torch.from_numpy(np.ascontiguousarray(node_raw_features, dtype=np.float32)).to(device)
        self.edge_raw_features = torch.from_numpy(np.ascontiguousarray(edge_raw_features, dtype=np.float32)).to(device)

        self.neighbor_sampler = neighbor_sampler
        self.node_feat_dim = self.node_raw_features.shape[1]
        self.edge_feat_dim = self.edge_raw_features.shape[1]
        self.time_feat_dim = time_feat_dim
        self.channel_embedding_dim = channel_embedding_dim
        self.patch_size = patch_size
        self.num_layers = num_layers
        self.num_heads = num_heads
        self.dropout = dropout
        self.max_input_sequence_length = max_input_sequence_length
        self.device = device

        # same construction order as the reference => same default-init RNG stream per seed
        self.time_encoder = TimeEncoder(time_dim=time_feat_dim)
        self.neighbor_co_occurrence_feat_dim = self.channel_embedding_dim
        self.neighbor_co_occurrence_encoder = NeighborCooccurrenceEncoder(self.neighbor_co_occurrence_feat_dim, device=self.device)
        self.projection_layer = nn.ModuleDict({
            "node": nn.Linear(self.patch_size * self.node_feat_dim, self.channel_embedding_dim, bias=True),
            "edge": nn.Linear(self.patch_size * self.edge_feat_dim, self.channel_embedding_dim, bias=True),
            "time": nn.Linear(self.patch_size * self.time_feat_dim, self.channel_embedding_dim, bias=True),
            "neighbor_co_occurrence": nn.Linear(self.patch_size * self.neighbor_co_occurrence_feat_dim, self.channel_embedding_dim, bias=True),
        })
        self.num_channels = 4
        self.transformers = nn.ModuleList([
            TransformerEncoder(attention_dim=self.num_channels * self.channel_embedding_dim, num_heads=self.num_heads, dropout=self.dropout)
            for _ in range(self.num_layers)])
        self.output_layer = nn.Linear(self.num_channels * self.channel_embedding_dim, self.node_feat_dim, bias=True)

        self._lib = _capi.load()           # fails loudly when the HIP library is missing
        self._cfg = _capi.DygformerConfig(self.node_feat_dim, self.edge_feat_dim, self.time_feat_dim, self.channel_embedding_dim,
                                          self.patch_size, self.num_layers, self.num_heads, self.max_input_sequence_length)
        self._packed: Optional[torch.Tensor] = None
        self._packed_key = None
        self._workspace: Dict[tuple, torch.Tensor] = {}
        self.impl = 0                      # 0 auto, 1 generic, 3 fused (see include/dygnn.h)

    # ---- reference API -------------------------------------------------------------------------
    def set_neighbor_sampler(self, neighbor_sampler: NeighborSampler):
        """models/DyGFormer.py:308-317."""
        self.neighbor_sampler = neighbor_sampler
        if self.neighbor_sampler.sample_neighbor_strategy in ["uniform", "time_interval_aware"]:
            assert self.neighbor_sampler.seed is not None
            self.neighbor_sampler.reset_random_state()

    def compute_src_dst_node_temporal_embeddings(self, src_node_ids, dst_node_ids, node_interact_times,
                                                 _taps: Optional[dict] = None, _group_size: int = 0, _pair_stride: int = 0
                                                 ) -> Tuple[torch.Tensor, torch.Tensor]:
        """models/DyGFormer.py:68-194.  ndarray (or already-resident device tensor) [B] int64, [B] int64,
        [B] float64 -> two float32 tensors [B, node_feat_dim] on the model's device."""
        dev = self._device()
        self._validate_ids(src_node_ids, dst_node_ids)
        src = self._to_dev(src_node_ids, torch.int64, dev)
        dst = self._to_dev(dst_node_ids, torch.int64, dev)
        tms = self._to_dev(node_interact_times, torch.float64, dev)
        B = src.numel()
        if not (dst.numel() == B and tms.numel() == B):
            raise AssertionError("src_node_ids, dst_node_ids and node_interact_times must have the same length")
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self._param_list())
        if (self.training or needs_grad) and B > 0 and _taps is None:
            if not self.training and not getattr(self, "_warned_eval_grad", False):
                # eval mode with autograd recording is a legitimate call (gradients without dropout) but ~8x slower than inference:
                # an evaluation loop that forgot torch.no_grad() should hear about it once
                import warnings
                warnings.warn("dyglib_amd.DyGFormer: eval-mode forward with autograd recording runs the training kernels (activations kept "
                              "for backward); wrap evaluation in torch.no_grad() for the fused inference kernel", RuntimeWarning, stacklevel=2)
                self._warned_eval_grad = True
            p_drop, seed = self._dropout_and_seed()
            seq_lens = self._seq_lens_side_stream(src_node_ids, dst_node_ids, node_interact_times, src, dst, tms, dev)
            return _TrainFunction.apply(self, src, dst, tms, p_drop, seed, seq_lens, *self._param_list())
        out_src = torch.empty((B, self.node_feat_dim), dtype=torch.float32, device=dev)
        out_dst = torch.empty_like(out_src)
        if B == 0:
            return out_src, out_dst
        weights, packed = self._packed_weights(dev)
        ws = self._workspace_for(B, dev)
        taps_struct = None
        if _taps is not None:
            taps_struct = self._make_taps(_taps, B, dev)
        elif getattr(self, "_kernel_events", None) is not None:
            # bench.py: time the fused kernel itself (events recorded by the library around its launch), not the whole call
            e0, e1 = self._kernel_events
            taps_struct = _capi.DygformerTaps()
            taps_struct.ev_kernel_start, taps_struct.ev_kernel_stop = e0.cuda_event, e1.cuda_event
        csr = self.neighbor_sampler.csr.on_device(dev)
        rc = self._lib.dygnn_dygformer_forward(
            C.byref(self._cfg), C.byref(weights), packed.data_ptr(), csr,
            self.node_raw_features.data_ptr(), self.edge_raw_features.data_ptr(),
            src.data_ptr(), dst.data_ptr(), tms.data_ptr(), B, int(_group_size), int(_pair_stride), out_src.data_ptr(), out_dst.data_ptr(),
            ws.data_ptr(), ws.numel(), C.byref(taps_struct) if taps_struct is not None else None,
            int(self.impl), _capi.current_stream_ptr())
        _capi.check(rc)
        return out_src, out_dst

    def compute_src_dst_node_temporal_embeddings_many(self, src_node_ids, dst_node_ids, node_interact_times, pos_neg_halves: bool = False):
        """Several independent reference calls in ONE launch: inputs are [N, B] (N calls of B pairs each; e.g. the
        positive and the negative call of a step, or N evaluation batches).  Row i of the result equals
        compute_src_dst_node_temporal_embeddings(src[i], dst[i], t[i]) bit for bit (every call keeps its own
        padded lengths), but the N*B pairs form one grid, which keeps all 256 CUs busy instead of B of them.
        pos_neg_halves=True (N even): calls N/2 .. N-1 are the NEGATIVE calls of calls 0 .. N/2-1 — same sources and times, other
        destinations (train_link_prediction.py:165-166) — so the kernel handles the two pairs of an edge together and gathers /
        projects the shared source side once; the rows stay bit-identical.
        Returns two float32 tensors [N, B, node_feat_dim]."""
        dev = self._device()
        self._validate_ids(src_node_ids, dst_node_ids)
        src = self._to_dev(src_node_ids, torch.int64, dev)
        dst = self._to_dev(dst_node_ids, torch.int64, dev)
        tms = self._to_dev(node_interact_times, torch.float64, dev)
        if src.dim() != 2 or src.shape != dst.shape or src.shape != tms.shape:
            raise AssertionError("expected three [N, B] arrays of equal shape")
        N, B = src.shape
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self._param_list())
        if (self.training or needs_grad) and N * B > 0:
            # training: the dense pass has ONE pair of padded lengths.  When every call of the set pads to the same lengths (the
            # usual case: some window of each call is full) the N calls are one pass over N*B pairs -- pairs never interact, so each
            # row equals the row of its own call and the parameter gradients are the sums of the per-call gradients, with half the
            # launches and no gradient-accumulation kernels; otherwise call by call.  Dropout masks are drawn per pass.
            lens = self._seq_lens_groups(src_node_ids, dst_node_ids, node_interact_times, src, dst, tms, dev)
            if all(l == lens[0] for l in lens):
                p_drop, seed = self._dropout_and_seed()
                a, b = _TrainFunction.apply(self, src.reshape(-1), dst.reshape(-1), tms.reshape(-1), p_drop, seed, lens[0], *self._param_list())
                return a.reshape(N, B, -1), b.reshape(N, B, -1)
            outs = [self.compute_src_dst_node_temporal_embeddings(src[i], dst[i], tms[i]) for i in range(N)]
            return torch.stack([o[0] for o in outs]), torch.stack([o[1] for o in outs])
        if pos_neg_halves and N % 2 != 0:
            raise AssertionError("pos_neg_halves needs an even number of calls: [positive calls ; negative calls]")
        a, b = self.compute_src_dst_node_temporal_embeddings(src.reshape(-1), dst.reshape(-1), tms.reshape(-1), _group_size=B,
                                                             _pair_stride=(N // 2) * B if pos_neg_halves else 0)
        return a.reshape(N, B, -1), b.reshape(N, B, -1)

    def _dropout_and_seed(self):
        for p in self._param_list():
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise _capi.DygnnError("parameters must be contiguous float32")
        seed = getattr(self, "_fixed_dropout_seed", None)             # tests pin the masks; normally torch.manual_seed governs them
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        return (float(self.dropout) if self.training else 0.0), seed

    def _seq_lens_groups(self, src_in, dst_in, t_in, src: torch.Tensor, dst: torch.Tensor, tms: torch.Tensor, dev):
        """(S_src, S_dst) of each of the N calls in [N, B] inputs, on the side stream (one synchronisation of IT, not of the main stream:
        the host keeps queueing this step while the GPU still works on the previous one).  Host inputs are uploaded a second time on the
        side stream; device inputs make the side stream wait for the main stream's copy of them."""
        side = getattr(self, "_side", None)
        if side is None or side.device != dev:
            side = self._side = torch.cuda.Stream(dev)
        L, P = self.max_input_sequence_length, self.patch_size
        N, B = src.shape
        host = not any(isinstance(x, torch.Tensor) for x in (src_in, dst_in, t_in))
        ev = None
        if not host:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            if host:
                src = torch.from_numpy(np.ascontiguousarray(src_in, dtype=np.int64)).to(dev)
                dst = torch.from_numpy(np.ascontiguousarray(dst_in, dtype=np.int64)).to(dev)
                tms = torch.from_numpy(np.ascontiguousarray(t_in, dtype=np.float64)).to(dev)
            else:
                side.wait_event(ev)
            hist = torch.empty(B, dtype=torch.int32, device=dev)
            end = torch.empty(B, dtype=torch.int64, device=dev)
            maxw = torch.zeros(N, 2, dtype=torch.int32, device=dev)
            csr = self.neighbor_sampler.csr.on_device(dev)
            for i in range(N):
                for half, nodes in enumerate((src[i], dst[i])):
                    _capi.check(self._lib.dygnn_window_lengths(csr, nodes.data_ptr(), tms[i].data_ptr(), B, L, hist.data_ptr(), end.data_ptr(),
                                                               maxw[i, half:half + 1].data_ptr(), side.cuda_stream))
            m = maxw.cpu()                               # synchronises the side stream only
        if not host:
            src.record_stream(side); dst.record_stream(side); tms.record_stream(side)
        return [tuple((int(v) + 1) + (P - (int(v) + 1) % P) % P for v in row) for row in m.tolist()]

    def _seq_lens_side_stream(self, src_in, dst_in, t_in, src, dst, tms, dev):
        """(S_src, S_dst) of this call, computed on a side stream so that the training forward does not have to synchronise the main
        stream (which would serialise the host's launch work with everything still queued on the GPU).  Host inputs are uploaded a
        second time on the side stream; device inputs make the side stream wait for the main stream's copy of them."""
        side = getattr(self, "_side", None)
        if side is None or side.device != dev:
            side = self._side = torch.cuda.Stream(dev)
        L, P = self.max_input_sequence_length, self.patch_size
        B = src.numel()
        host = not any(isinstance(x, torch.Tensor) for x in (src_in, dst_in, t_in))
        ev = None
        if not host:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            if host:
                nodes = torch.from_numpy(np.concatenate([np.asarray(src_in, dtype=np.int64), np.asarray(dst_in, dtype=np.int64)])).to(dev)
                times = torch.from_numpy(np.concatenate([np.asarray(t_in, dtype=np.float64)] * 2)).to(dev)
            else:
                side.wait_event(ev)
                nodes, times = torch.cat([src, dst]), torch.cat([tms, tms])
            hist = torch.empty(2 * B, dtype=torch.int32, device=dev)
            end = torch.empty(2 * B, dtype=torch.int64, device=dev)
            maxw = torch.zeros(2, dtype=torch.int32, device=dev)
            csr = self.neighbor_sampler.csr.on_device(dev)
            for half in (0, 1):
                sl = slice(half * B, (half + 1) * B)
                _capi.check(self._lib.dygnn_window_lengths(csr, nodes[sl].data_ptr(), times[sl].data_ptr(), B, L, hist[sl].data_ptr(),
                                                           end[sl].data_ptr(), maxw[half:half + 1].data_ptr(), side.cuda_stream))
            m = maxw.cpu()                               # synchronises the side stream only
        S = [int(v) + 1 for v in m.tolist()]
        return tuple(s_ + (P - s_ % P) % P for s_ in S)

    # ---- plumbing ------------------------------------------------------------------------------
    def _validate_ids(self, *id_arrays) -> None:
        """The reference trusts ids and fails with IndexError (list / tensor indexing); here host inputs are range-checked per call
        (two min/max over B ids) and the graph against the feature tables once per sampler, so no kernel can index outside a table."""
        csr = self.neighbor_sampler.csr
        if getattr(self, "_validated_csr", None) is not csr:
            csr.check_tables(self.node_raw_features.shape[0], self.edge_raw_features.shape[0])
            self._validated_csr = csr
        for ids in id_arrays:
            csr.check_query_ids(ids, limit=self.node_raw_features.shape[0])

    def invalidate_packed(self) -> None:
        """Drop the kernel-ready weight copy.  It is re-packed automatically when a parameter's version counter or address changes
        (optimizer steps, load_state_dict, .to()), on every train()/eval() switch and by this call; writes through `p.data` do not
        bump the version counter, so code that edits weights that way (custom init, EMA) calls this afterwards."""
        self._packed_key = None

    def train(self, mode: bool = True):
        self._packed_key = None
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        self._packed_key = None
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._packed_key = None
        return super().load_state_dict(*args, **kwargs)

    def _device(self) -> torch.device:
        dev = self.output_layer.weight.device
        if dev.type != "cuda":
            raise _capi.DygnnError("dyglib_amd.DyGFormer runs on an MI355X only: move the model to a GPU "
                                   "(convert_to_gpu / .to('cuda')); there is no CPU fallback")
        if self.node_raw_features.device != dev:      # the reference places the tables at construction
            self.node_raw_features = self.node_raw_features.to(dev)
            self.edge_raw_features = self.edge_raw_features.to(dev)
        return dev

    @staticmethod
    def _to_dev(x, dtype, dev) -> torch.Tensor:
        if isinstance(x, torch.Tensor):
            return x.to(device=dev, dtype=dtype).contiguous()
        return torch.from_numpy(np.ascontiguousarray(x, dtype={torch.int64: np.int64, torch.float64: np.float64}[dtype])).to(dev, non_blocking=True)

    def _weights_struct(self, replace: Optional[dict] = None) -> "_capi.DygformerWeights":
        """ctypes view of the parameters; `replace` maps parameter names to other tensors of the same shapes (the gradient
        buffers of the backward pass)."""
        w = _capi.DygformerWeights()
        if replace is None:
            p = lambda t: t.data_ptr()
        elif isinstance(replace, dict):
            by_id = {id(t): replace[n] for n, t in self.named_parameters()}
            p = lambda t: by_id[id(t)].data_ptr()
        else:                           # a list aligned with self._param_list() (the backward pass: no module traversal per step)
            index = self.__dict__.get("_pindex")
            if index is None:
                index = self.__dict__["_pindex"] = {id(t): i for i, t in enumerate(self._param_list())}
            p = lambda t: replace[index[id(t)]].data_ptr()
        enc = self.neighbor_co_occurrence_encoder.neighbor_co_occurrence_encode_layer
        w.time_w, w.time_b = p(self.time_encoder.w.weight), p(self.time_encoder.w.bias)
        w.cooc_w0, w.cooc_b0, w.cooc_w1, w.cooc_b1 = p(enc[0].weight), p(enc[0].bias), p(enc[2].weight), p(enc[2].bias)
        pl = self.projection_layer
        w.proj_node_w, w.proj_node_b = p(pl["node"].weight), p(pl["node"].bias)
        w.proj_edge_w, w.proj_edge_b = p(pl["edge"].weight), p(pl["edge"].bias)
        w.proj_time_w, w.proj_time_b = p(pl["time"].weight), p(pl["time"].bias)
        w.proj_cooc_w, w.proj_cooc_b = p(pl["neighbor_co_occurrence"].weight), p(pl["neighbor_co_occurrence"].bias)
        for l, tr in enumerate(self.transformers):
            L = w.layers[l]
            mha = tr.multi_head_attention
            L.in_proj_weight, L.in_proj_bias = p(mha.in_proj_weight), p(mha.in_proj_bias)
            L.out_proj_weight, L.out_proj_bias = p(mha.out_proj.weight), p(mha.out_proj.bias)
            L.ffn0_weight, L.ffn0_bias = p(tr.linear_layers[0].weight), p(tr.linear_layers[0].bias)
            L.ffn1_weight, L.ffn1_bias = p(tr.linear_layers[1].weight), p(tr.linear_layers[1].bias)
            L.norm0_weight, L.norm0_bias = p(tr.norm_layers[0].weight), p(tr.norm_layers[0].bias)
            L.norm1_weight, L.norm1_bias = p(tr.norm_layers[1].weight), p(tr.norm_layers[1].bias)
        w.output_w, w.output_b = p(self.output_layer.weight), p(self.output_layer.bias)
        return w

    def _param_list(self):
        plist = self.__dict__.get("_plist")
        if plist is None:
            plist = self.__dict__["_plist"] = list(self.parameters())          # parameters are never added after construction
        return plist

    def _packed_weights(self, dev, training: bool = False):
        """(ctypes view of the parameters, kernel-ready weight copy); both rebuilt whenever any parameter was written (optimizer
        step, load_state_dict) — detected through the tensors' version counters and addresses."""
        plist = self._param_list()
        key = tuple([(p.data_ptr(), p._version) for p in plist]) + (str(dev),)
        if self._packed is not None and self._packed_key == key and not training and not self.__dict__.get("_packed_complete", True):
            # the training path refreshed the fragment streams only: bring the inference sections up to date too
            _capi.check(self._lib.dygnn_dygformer_repack(C.byref(self._cfg), C.byref(self._weights_cached), self._packed.data_ptr(), self._packed.numel(), 0,
                                                         _capi.current_stream_ptr()))
            self.__dict__["_packed_complete"] = True
        if self._packed is None or self._packed_key != key:
            for p in plist:
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise _capi.DygnnError("parameters must be contiguous float32")
            weights = self._weights_cached = self._weights_struct()
            nbytes = self._lib.dygnn_dygformer_packed_bytes(C.byref(self._cfg))
            if nbytes == 0:
                _capi.check(-1)
            fresh = self._packed is None or self._packed.numel() != nbytes or self._packed.device != dev
            if fresh:
                self._packed = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
            ptrs = tuple(k[0] for k in key[:-1]) + (self._packed.data_ptr(),)
            # an optimizer step changes the values, not the addresses: the fragment descriptors already in the buffer stay valid and the
            # refresh is kernel launches only (no host synchronisation inside the training loop)
            repack = not fresh and self.__dict__.get("_packed_ptrs") == ptrs
            self.__dict__["_packed_ptrs"] = None
            if repack:
                _capi.check(self._lib.dygnn_dygformer_repack(C.byref(self._cfg), C.byref(weights), self._packed.data_ptr(), nbytes, 1 if training else 0,
                                                             _capi.current_stream_ptr()))
                self.__dict__["_packed_complete"] = not training
            else:
                _capi.check(self._lib.dygnn_dygformer_pack(C.byref(self._cfg), C.byref(weights), self._packed.data_ptr(), nbytes, _capi.current_stream_ptr()))
                self.__dict__["_packed_complete"] = True
            self.__dict__["_packed_ptrs"] = ptrs
            self._packed_key = key
        return self._weights_cached, self._packed

    def _workspace_for(self, B: int, dev) -> torch.Tensor:
        # one workspace per (batch size, stream): calls issued on different HIP streams may overlap
        key = (B, int(self.impl), torch.cuda.current_stream(dev).cuda_stream)
        ws = self._workspace.get(key)
        if ws is None or ws.device != dev:
            nbytes = self._lib.dygnn_dygformer_workspace_bytes_for(C.byref(self._cfg), B, int(self.impl))
            ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
            if len(self._workspace) > 16:
                self._workspace.clear()
            self._workspace[key] = ws
        return ws

    def _make_taps(self, taps: dict, B: int, dev) -> "_capi.DygformerTaps":
        P, L, D = self.patch_size, self.max_input_sequence_length, self.num_channels * self.channel_embedding_dim
        t_max = 2 * ((L + P - 1) // P)
        taps["seq_lens"] = torch.zeros(2, dtype=torch.int32, device=dev)
        taps["encoder_input"] = torch.zeros((B, t_max, D), dtype=torch.float32, device=dev)
        taps["layer_outputs"] = [torch.zeros((B, t_max, D), dtype=torch.float32, device=dev) for _ in range(self.num_layers)]
        s = _capi.DygformerTaps()
        s.seq_lens = taps["seq_lens"].data_ptr()
        s.encoder_input = taps["encoder_input"].data_ptr()
        for l in range(self.num_layers):
            s.layer_out[l] = taps["layer_outputs"][l].data_ptr()
        if taps.get("want_phase_cycles"):        # only the -DDYGNN_STAMPS diagnostic build writes them
            taps["phase_cycles"] = torch.zeros((4, 8, 32), dtype=torch.int64, device=dev)
            s.phase_cycles = taps["phase_cycles"].data_ptr()
        return s
