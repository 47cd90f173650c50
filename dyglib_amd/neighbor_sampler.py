"""Drop-in for the reference `NeighborSampler` / `get_neighbor_sampler`
(utils/utils.py:71-302) backed by the gfx950 kernels behind include/dygnn.h.

Same constructor, method names, argument meaning, return dtypes and error behaviour as the
reference.  `recent` sampling and `get_all_first_hop_neighbors` run entirely on the GPU.  `uniform`
and `time_interval_aware` (utils/utils.py:176-199, :112-128) must consume the sampler's numpy
RandomState in batch-row order to stay bit-identical to the reference, so for those two the GPU
does the time searches and the final gather while the draws, the float32 time lookup and the
(unstable) argsort of every row are replayed on the host with numpy itself.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import ctypes as C
import os

import numpy as np
import torch

from . import _capi
from .temporal_csr import TemporalCSR


def _default_device() -> torch.device:
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


class NeighborSampler:

    def __init__(self, adj_list: Optional[list] = None, sample_neighbor_strategy: str = "uniform",
                 time_scaling_factor: float = 0.0, seed: Optional[int] = None, *, csr: Optional[TemporalCSR] = None,
                 device=None):
        """Reference signature: NeighborSampler(adj_list, sample_neighbor_strategy, time_scaling_factor, seed)
        (utils/utils.py:73).  `csr=` is the fast path used by get_neighbor_sampler."""
        self.sample_neighbor_strategy = sample_neighbor_strategy
        self.seed = seed
        self.time_scaling_factor = time_scaling_factor
        if csr is None:
            if adj_list is None:
                raise TypeError("NeighborSampler needs adj_list or csr")
            csr = TemporalCSR.from_adj_list(adj_list)
        self.csr = csr
        self.device = torch.device(device) if device is not None else _default_device()
        self._lib = _capi.load()
        self._host64 = None
        if self.sample_neighbor_strategy == "time_interval_aware":
            # utils/utils.py:105-107: per node, probabilities over its whole time-sorted history
            self.nodes_neighbor_sampled_probabilities = [
                self.compute_sampled_probabilities(csr.ts[csr.indptr[v]:csr.indptr[v + 1]]) for v in range(csr.num_nodes)]
        if self.seed is not None:
            self.random_state = np.random.RandomState(self.seed)          # utils/utils.py:109-110

    def compute_sampled_probabilities(self, node_neighbor_times: np.ndarray):
        """utils/utils.py:112-128, the same numpy float64 operations in the same order."""
        if len(node_neighbor_times) == 0:
            return np.array([])
        node_neighbor_times = node_neighbor_times - np.max(node_neighbor_times)
        exp_node_neighbor_times = np.exp(self.time_scaling_factor * node_neighbor_times)
        with np.errstate(divide="ignore", invalid="ignore"):
            sampled_probabilities = exp_node_neighbor_times / np.cumsum(exp_node_neighbor_times)
        sampled_probabilities[np.isnan(sampled_probabilities)] = -1e10
        return sampled_probabilities

    # ---- helpers -------------------------------------------------------------------------------
    def _queries_to_device(self, node_ids, node_interact_times) -> Tuple[torch.Tensor, torch.Tensor]:
        self.csr.check_query_ids(node_ids)              # IndexError like the reference's list index (utils/utils.py:139)
        nodes = torch.as_tensor(np.ascontiguousarray(node_ids, dtype=np.int64)).to(self.device, non_blocking=True)
        # float32 query times (TGAT second hop, models/TGAT.py:107-110) widen exactly to float64
        times = torch.as_tensor(np.ascontiguousarray(node_interact_times, dtype=np.float64)).to(self.device, non_blocking=True)
        if nodes.shape != times.shape or nodes.dim() != 1:
            raise AssertionError("node_ids and node_interact_times must be 1-D arrays of equal length")
        return nodes, times

    def _host_rows64(self):
        if self._host64 is None:
            self._host64 = (self.csr.nbr.astype(np.int64), self.csr.eid.astype(np.int64))
        return self._host64

    def _check_strategy(self):
        if self.sample_neighbor_strategy not in ("recent", "uniform", "time_interval_aware"):
            # utils/utils.py:211
            raise ValueError(f"Not implemented error for sample_neighbor_strategy {self.sample_neighbor_strategy}!")

    def _require_recent(self):
        """The fused TGAT / TGN forwards sample on the device, which only the stateless `recent` strategy allows."""
        self._check_strategy()
        if self.sample_neighbor_strategy != "recent":
            raise NotImplementedError(
                f"the fused TGAT/TGN forward samples on the device and supports sample_neighbor_strategy 'recent' only; "
                f"'{self.sample_neighbor_strategy}' is available through NeighborSampler.get_historical_neighbors")

    def _draw_host(self, node_ids_host: np.ndarray, hist_len_host: np.ndarray, k: int) -> np.ndarray:
        """The reference's per-row draw (utils/utils.py:176-199) replayed with numpy: returns sel [n,k] int32 =
        positions inside each node's row in FINAL order (after the argsort of the float32 times), -1 = no history."""
        tia = self.sample_neighbor_strategy == "time_interval_aware"
        indptr, ts = self.csr.indptr, self.csr.ts
        if not tia and self.seed is not None and os.environ.get("DYGNN_SAMPLER_PYTHON_DRAWS") != "1":
            # `uniform` on a seeded sampler: ALL rows' draws in one library call on the RandomState's own MT19937 state
            # (dygnn_mt19937_choice_rows_host = RandomState.choice(a=cnt, size=k) row by row), then the reference's per-row steps as array
            # operations: float32 times of the drawn positions, argsort along the row (numpy's own argsort: the same tie order as the
            # reference's per-row call, utils/utils.py:196), positions in that order.  tests/test_sampling_strategies.py pins it to the
            # Python loop below and to the reference's fixtures.
            nodes = np.ascontiguousarray(node_ids_host, dtype=np.int64)
            cnt = np.ascontiguousarray(hist_len_host, dtype=np.int32)
            st = self.random_state.get_state(legacy=True)
            key, pos = np.ascontiguousarray(st[1], dtype=np.uint32).copy(), C.c_int32(int(st[2]))
            sampled = np.empty((len(nodes), k), dtype=np.int32)
            _capi.check(self._lib.dygnn_mt19937_choice_rows_host(key.ctypes.data, C.byref(pos), cnt.ctypes.data, len(nodes), k, sampled.ctypes.data))
            self.random_state.set_state(("MT19937", key, int(pos.value), st[3], st[4]))
            t32 = ts[indptr[nodes][:, None] + sampled].astype(np.float32)                   # utils/utils.py:192 (rows without history: any valid position)
            sel = np.take_along_axis(sampled, t32.argsort(axis=1), axis=1)                   # utils/utils.py:196-199
            sel[cnt <= 0] = -1
            return np.ascontiguousarray(sel, dtype=np.int32)
        sel = np.full((len(node_ids_host), k), -1, dtype=np.int32)
        rng = self.random_state if self.seed is not None else np.random
        for idx, (node, cnt) in enumerate(zip(node_ids_host.tolist(), hist_len_host.tolist())):
            if cnt <= 0:
                continue
            p = None
            if tia:
                probs = self.nodes_neighbor_sampled_probabilities[node][:cnt]
                p = torch.softmax(torch.from_numpy(probs).float(), dim=0).numpy()          # utils/utils.py:184
            sampled = rng.choice(a=cnt, size=k, p=p)                                        # utils/utils.py:186-188
            t32 = ts[indptr[node] + sampled].astype(np.float32)                             # utils/utils.py:192 (float32 row)
            sel[idx] = sampled[t32.argsort()]                                               # utils/utils.py:196-199
        return sel

    # ---- device-resident API (no host round trip) ----------------------------------------------
    def hist_len_device(self, nodes: torch.Tensor, times: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """find_neighbors_before for device query tensors -> (hist_len int32 [n], end_pos int64 [n])."""
        n = nodes.numel()
        hist = torch.empty(n, dtype=torch.int32, device=self.device)
        end = torch.empty(n, dtype=torch.int64, device=self.device)
        rc = self._lib.dygnn_find_neighbors_before(self.csr.on_device(self.device), nodes.data_ptr(), times.data_ptr(), n,
                                                   hist.data_ptr(), end.data_ptr(), _capi.current_stream_ptr())
        _capi.check(rc)
        return hist, end

    def get_historical_neighbors_device(self, nodes: torch.Tensor, times: torch.Tensor, num_neighbors: int = 20):
        """Sampling on device tensors; returns device tensors ([n,k] int64, int64, float32).  `recent` stays on the
        device; the random strategies make one host round trip (hist_len down, selected positions up)."""
        self._check_strategy()
        n, k = nodes.numel(), int(num_neighbors)
        out_n = torch.empty((n, max(k, 0)), dtype=torch.int64, device=self.device)
        out_e = torch.empty_like(out_n)
        out_t = torch.empty((n, max(k, 0)), dtype=torch.float32, device=self.device)
        if self.sample_neighbor_strategy != "recent":
            assert k > 0, "Number of sampled neighbors for each node should be greater than 0!"
            hist, _ = self.hist_len_device(nodes, times)
            sel = self._draw_host(nodes.cpu().numpy(), hist.cpu().numpy(), k)
            sel_d = torch.from_numpy(sel).to(self.device)
            rc = self._lib.dygnn_gather_selected(self.csr.on_device(self.device), nodes.data_ptr(), sel_d.data_ptr(), n, k,
                                                 out_n.data_ptr(), out_e.data_ptr(), out_t.data_ptr(), _capi.current_stream_ptr())
            _capi.check(rc)
            return out_n, out_e, out_t
        rc = self._lib.dygnn_sample_recent(self.csr.on_device(self.device), nodes.data_ptr(), times.data_ptr(), n, k,
                                           out_n.data_ptr(), out_e.data_ptr(), out_t.data_ptr(), _capi.current_stream_ptr())
        _capi.check(rc)                                    # k <= 0 -> AssertionError, utils/utils.py:157
        return out_n, out_e, out_t

    # ---- reference API (numpy in, numpy out) ---------------------------------------------------
    def find_neighbors_before(self, node_id: int, interact_time: float, return_sampled_probabilities: bool = False):
        """utils/utils.py:130-147 for a single query (views into the host copy of the CSR)."""
        nodes, times = self._queries_to_device(np.array([node_id]), np.array([interact_time]))
        hist, _ = self.hist_len_device(nodes, times)
        i = int(hist.cpu()[0])
        a = int(self.csr.indptr[node_id])
        n64, e64 = self._host_rows64()
        probs = self.nodes_neighbor_sampled_probabilities[node_id][:i] if return_sampled_probabilities else None
        return n64[a:a + i], e64[a:a + i], self.csr.ts[a:a + i], probs

    def get_historical_neighbors(self, node_ids: np.ndarray, node_interact_times: np.ndarray, num_neighbors: int = 20):
        """utils/utils.py:149-214: ndarrays [n,k] int64, int64, float32."""
        assert num_neighbors > 0, "Number of sampled neighbors for each node should be greater than 0!"
        nodes, times = self._queries_to_device(node_ids, node_interact_times)
        n, e, t = self.get_historical_neighbors_device(nodes, times, num_neighbors)
        return n.cpu().numpy(), e.cpu().numpy(), t.cpu().numpy()

    def get_multi_hop_neighbors(self, num_hops: int, node_ids: np.ndarray, node_interact_times: np.ndarray,
                                num_neighbors: int = 20):
        """utils/utils.py:216-252; hop h>1 queries use the float32 times returned by hop h-1."""
        assert num_hops > 0, "Number of sampled hops should be greater than 0!"
        nodes, times = self._queries_to_device(node_ids, node_interact_times)
        B = nodes.numel()
        n, e, t = self.get_historical_neighbors_device(nodes, times, num_neighbors)
        ids_l, eids_l, ts_l = [n], [e], [t]
        for _ in range(1, num_hops):
            n, e, t = self.get_historical_neighbors_device(ids_l[-1].reshape(-1), ts_l[-1].reshape(-1).double(), num_neighbors)
            ids_l.append(n.reshape(B, -1)), eids_l.append(e.reshape(B, -1)), ts_l.append(t.reshape(B, -1))
        to_np = lambda xs: [x.cpu().numpy() for x in xs]
        return to_np(ids_l), to_np(eids_l), to_np(ts_l)

    def get_all_first_hop_neighbors(self, node_ids: np.ndarray, node_interact_times: np.ndarray):
        """utils/utils.py:254-273: three lists of variable-length arrays (int64, int64, float64).  The
        searches run on the GPU; the lists are views of the host copy of the CSR."""
        nodes, times = self._queries_to_device(node_ids, node_interact_times)
        hist, _ = self.hist_len_device(nodes, times)
        hist = hist.cpu().numpy()
        n64, e64 = self._host_rows64()
        ids_l: List[np.ndarray] = []
        eids_l: List[np.ndarray] = []
        ts_l: List[np.ndarray] = []
        starts = self.csr.indptr[np.asarray(node_ids, dtype=np.int64)]
        for a, i in zip(starts.tolist(), hist.tolist()):
            ids_l.append(n64[a:a + i]), eids_l.append(e64[a:a + i]), ts_l.append(self.csr.ts[a:a + i])
        return ids_l, eids_l, ts_l

    def padded_windows(self, node_ids: np.ndarray, node_interact_times: np.ndarray, patch_size: int = 1,
                       max_input_sequence_length: int = 256):
        """get_all_first_hop_neighbors + DyGFormer.pad_sequences (models/DyGFormer.py:196-245) in two
        launches: returns ndarrays [n,S] int64, int64, float32."""
        nodes, times = self._queries_to_device(node_ids, node_interact_times)
        ids, eids, ts = self.padded_windows_device(nodes, times, patch_size, max_input_sequence_length)
        return ids.cpu().numpy(), eids.cpu().numpy(), ts.cpu().numpy()

    def padded_windows_device(self, nodes: torch.Tensor, times: torch.Tensor, patch_size: int, max_input_sequence_length: int):
        n = nodes.numel()
        L = int(max_input_sequence_length)
        hist = torch.empty(n, dtype=torch.int32, device=self.device)
        end = torch.empty(n, dtype=torch.int64, device=self.device)
        maxw = torch.zeros(1, dtype=torch.int32, device=self.device)
        csr = self.csr.on_device(self.device)
        s = _capi.current_stream_ptr()
        _capi.check(self._lib.dygnn_window_lengths(csr, nodes.data_ptr(), times.data_ptr(), n, L, hist.data_ptr(),
                                                   end.data_ptr(), maxw.data_ptr(), s))
        S = int(maxw.item()) + 1                                   # the one host sync: S sizes the output
        if S % patch_size != 0:
            S += patch_size - S % patch_size
        ids = torch.empty((n, S), dtype=torch.int64, device=self.device)
        eids = torch.empty_like(ids)
        ts = torch.empty((n, S), dtype=torch.float32, device=self.device)
        _capi.check(self._lib.dygnn_window_fill(csr, nodes.data_ptr(), times.data_ptr(), n, L, S, hist.data_ptr(),
                                                end.data_ptr(), ids.data_ptr(), eids.data_ptr(), ts.data_ptr(), s))
        return ids, eids, ts

    def reset_random_state(self):
        """utils/utils.py:275-280."""
        self.random_state = np.random.RandomState(self.seed)


def count_nodes_appearances(src_padded_nodes_neighbor_ids, dst_padded_nodes_neighbor_ids, device=None):
    """NeighborCooccurrenceEncoder.count_nodes_appearances (models/DyGFormer.py:337-393) on the GPU:
    ndarray/tensor int64 [B,S_s], [B,S_d] -> float32 tensors [B,S_s,2], [B,S_d,2] on the device."""
    lib = _capi.load()
    device = torch.device(device) if device is not None else _default_device()
    s = torch.as_tensor(src_padded_nodes_neighbor_ids).to(device=device, dtype=torch.int64).contiguous()
    d = torch.as_tensor(dst_padded_nodes_neighbor_ids).to(device=device, dtype=torch.int64).contiguous()
    assert s.dim() == 2 and d.dim() == 2 and s.shape[0] == d.shape[0]
    cs = torch.empty(s.shape + (2,), dtype=torch.float32, device=device)
    cd = torch.empty(d.shape + (2,), dtype=torch.float32, device=device)
    _capi.check(lib.dygnn_cooccurrence(s.data_ptr(), d.data_ptr(), s.shape[0], s.shape[1], d.shape[1], cs.data_ptr(),
                                       cd.data_ptr(), _capi.current_stream_ptr()))
    return cs, cd


def get_neighbor_sampler(data, sample_neighbor_strategy: str = "uniform", time_scaling_factor: float = 0.0,
                         seed: Optional[int] = None, device=None) -> NeighborSampler:
    """utils/utils.py:283-302: `data` is anything with src_node_ids / dst_node_ids / edge_ids /
    node_interact_times arrays (the reference `Data`, or dyglib_amd.synthetic.InteractionData)."""
    csr = TemporalCSR.from_interactions(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    return NeighborSampler(None, sample_neighbor_strategy=sample_neighbor_strategy, time_scaling_factor=time_scaling_factor,
                           seed=seed, csr=csr, device=device)
