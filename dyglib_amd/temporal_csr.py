"""Time-sorted CSR adjacency: the HBM-resident replacement of the reference's per-node Python
lists (utils/utils.py:85-103).  Built once on the host by the C++ builder behind
`dygnn_csr_build_host`, uploaded once per GPU, immutable afterwards.

HBM layout (SoA, one contiguous array per field so a row's tail window is three coalesced reads):
    indptr int64 [N+1] | nbr int32 [2E] | eid int32 [2E] | ts float64 [2E]
16 bytes per adjacency entry instead of the 24 of an (int64,int64,float64) layout; timestamps stay
float64 because the strictly-earlier test (utils/utils.py:139-141) is on the stored float64 value.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _capi


class TemporalCSR:
    def __init__(self, indptr: np.ndarray, nbr: np.ndarray, eid: np.ndarray, ts: np.ndarray):
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        self.nbr = np.ascontiguousarray(nbr, dtype=np.int32)
        self.eid = np.ascontiguousarray(eid, dtype=np.int32)
        self.ts = np.ascontiguousarray(ts, dtype=np.float64)
        self.num_nodes = len(self.indptr) - 1
        self.num_entries = len(self.nbr)
        self._dev = {}          # device -> (tensors, ctypes struct)
        self._max_ids = None    # (largest neighbour id, largest edge id), computed on first use

    # ---- validation (the reference trusts ids: numpy / list indexing raises IndexError there) -----------------
    def max_ids(self):
        if self._max_ids is None:
            self._max_ids = (int(self.nbr.max()) if self.num_entries else 0, int(self.eid.max()) if self.num_entries else 0)
        return self._max_ids

    def check_tables(self, num_node_rows: int, num_edge_rows: int) -> None:
        """Every id the kernels can read through this graph indexes inside the feature tables: the rows of the query ids
        (< num_nodes, see check_query_ids), of the CSR neighbours and of the CSR edge ids.  Called once per (sampler, model)."""
        max_nbr, max_eid = self.max_ids()
        if self.num_entries and (int(self.nbr.min()) < 0 or int(self.eid.min()) < 0):
            raise IndexError("negative node / edge id in the adjacency")
        if max_nbr >= num_node_rows:
            raise IndexError(f"index {max_nbr} is out of bounds for node_raw_features with {num_node_rows} rows")
        if max_eid >= num_edge_rows:
            raise IndexError(f"index {max_eid} is out of bounds for edge_raw_features with {num_edge_rows} rows")

    def check_query_ids(self, ids, limit: Optional[int] = None, what: str = "node id") -> None:
        """IndexError for host (numpy / list) query ids outside [0, min(num_nodes, limit)) — the reference's behaviour
        (utils/utils.py:139 list index, models/DyGFormer.py:259 tensor index).  Device tensors are not inspected (that would be a
        synchronisation): the kernels treat such an id as the padding node."""
        if isinstance(ids, torch.Tensor):
            return
        a = np.asarray(ids)
        if a.size == 0:
            return
        hi = self.num_nodes if limit is None else min(self.num_nodes, int(limit))
        lo_v, hi_v = int(a.min()), int(a.max())
        if lo_v < 0 or hi_v >= hi:
            bad = lo_v if lo_v < 0 else hi_v
            raise IndexError(f"{what} {bad} is out of bounds for a graph / feature table with {hi} rows")

    # ---- construction -----------------------------------------------------------------------
    @classmethod
    def from_interactions(cls, src: np.ndarray, dst: np.ndarray, eid: np.ndarray, ts: np.ndarray,
                          num_nodes: Optional[int] = None) -> "TemporalCSR":
        """get_neighbor_sampler's graph (utils/utils.py:293-300): undirected, rows indexed by node id,
        num_nodes = max id + 1 (row 0 = padding node)."""
        lib = _capi.load()
        src = np.ascontiguousarray(src, dtype=np.int64)
        dst = np.ascontiguousarray(dst, dtype=np.int64)
        eid = np.ascontiguousarray(eid, dtype=np.int64)
        ts = np.ascontiguousarray(ts, dtype=np.float64)
        E = len(src)
        if not (len(dst) == E and len(eid) == E and len(ts) == E):
            raise AssertionError("src/dst/edge id/time arrays must have the same length")
        if num_nodes is None:
            num_nodes = int(max(src.max(), dst.max())) + 1 if E else 1
        indptr = np.empty(num_nodes + 1, dtype=np.int64)
        nbr = np.empty(2 * E, dtype=np.int32)
        eo = np.empty(2 * E, dtype=np.int32)
        to = np.empty(2 * E, dtype=np.float64)
        rc = lib.dygnn_csr_build_host(E, src.ctypes.data, dst.ctypes.data, eid.ctypes.data, ts.ctypes.data, num_nodes,
                                      indptr.ctypes.data, nbr.ctypes.data, eo.ctypes.data, to.ctypes.data)
        _capi.check(rc, invalid_exc=IndexError)
        return cls(indptr, nbr, eo, to)

    @classmethod
    def from_adj_list(cls, adj_list) -> "TemporalCSR":
        """The reference constructor's input (utils/utils.py:73-103): adj_list[node] = list of
        (neighbor id, edge id, timestamp) tuples in insertion order; stable sort by time per node."""
        counts = np.fromiter((len(x) for x in adj_list), dtype=np.int64, count=len(adj_list))
        indptr = np.zeros(len(adj_list) + 1, dtype=np.int64)
        np.cumsum(counts, out=indptr[1:])
        total = int(indptr[-1])
        nbr = np.empty(total, dtype=np.int32)
        eid = np.empty(total, dtype=np.int32)
        ts = np.empty(total, dtype=np.float64)
        for node, row in enumerate(adj_list):
            if not row:
                continue
            a = indptr[node]
            arr_t = np.array([x[2] for x in row], dtype=np.float64)
            order = np.argsort(arr_t, kind="stable")
            nbr[a:a + len(row)] = np.array([x[0] for x in row], dtype=np.int64)[order]
            eid[a:a + len(row)] = np.array([x[1] for x in row], dtype=np.int64)[order]
            ts[a:a + len(row)] = arr_t[order]
        return cls(indptr, nbr, eid, ts)

    # ---- device residency ---------------------------------------------------------------------
    def on_device(self, device) -> "_capi.Csr":
        """ctypes view of the CSR resident on `device` (uploaded on first use, then cached)."""
        device = torch.device(device)
        key = str(device)
        if key not in self._dev:
            if device.type != "cuda":
                raise _capi.DygnnError("TemporalCSR.on_device needs a GPU device (the kernels are gfx950 HIP only)")
            t = dict(indptr=torch.from_numpy(self.indptr).to(device), nbr=torch.from_numpy(self.nbr).to(device),
                     eid=torch.from_numpy(self.eid).to(device), ts=torch.from_numpy(self.ts).to(device))
            s = _capi.Csr(self.num_nodes, self.num_entries, t["indptr"].data_ptr(),
                          t["nbr"].data_ptr() if self.num_entries else None,
                          t["eid"].data_ptr() if self.num_entries else None,
                          t["ts"].data_ptr() if self.num_entries else None)
            self._dev[key] = (t, s)
        return self._dev[key][1]

    def nbytes(self) -> int:
        return self.indptr.nbytes + self.nbr.nbytes + self.eid.nbytes + self.ts.nbytes
