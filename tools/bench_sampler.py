#!/usr/bin/env python3
"""Stage-level measurement of the neighbour-lookup kernels (sampler.hip, cooccurrence.hip) through the C ABI:
queries/s and algorithmic HBM GB/s (SURVEY §8d byte model: 8*ceil(log2(deg+1)) probe bytes + 16 B per window entry of this CSR + outputs) against the 8 TB/s HBM3E peak, on the Wikipedia- and
Reddit-shaped synthetic graphs.  The reference's get_historical_neighbors runs 149 k queries/s on the CPU.
    python tools/bench_sampler.py [--queries 2000000]
Prints one JSON line per (graph, kernel)."""
import argparse
import json
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dyglib_amd import _capi, get_neighbor_sampler, count_nodes_appearances, synthetic as syn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--queries", type=int, default=2_000_000)
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda:0")
HBM_PEAK = 8.0e12


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


for gname, (U, I, E) in {"wikipedia": (8227, 1000, 157474), "reddit": (10000, 984, 672447)}.items():
    data, _, _ = syn.make_bipartite_graph(U, I, E, seed=0)
    s = get_neighbor_sampler(data, "recent", seed=1, device=dev)
    rs = np.random.RandomState(1)
    # queries = (endpoint, time) of random interactions from the evaluation span (last 30 %), both endpoints
    idx = rs.randint(int(0.7 * E), E, size=args.queries // 2)
    nodes_h = np.concatenate([data.src_node_ids[idx], data.dst_node_ids[idx]])
    times_h = np.concatenate([data.node_interact_times[idx], data.node_interact_times[idx]])
    nodes = torch.from_numpy(nodes_h).to(dev)
    times = torch.from_numpy(times_h).to(dev)
    n = nodes.numel()
    deg = (s.csr.indptr[nodes_h + 1] - s.csr.indptr[nodes_h]).astype(np.float64)
    probe_bytes = 8 * np.ceil(np.log2(deg + 1))                              # binary-search probes, 8 B each
    lib = s._lib
    csr = s.csr.on_device(dev)
    st = _capi.current_stream_ptr()
    for k in (20, 10):
        on = torch.empty((n, k), dtype=torch.int64, device=dev)
        oe = torch.empty_like(on)
        ot = torch.empty((n, k), dtype=torch.float32, device=dev)
        sec = timed(lambda: _capi.check(lib.dygnn_sample_recent(csr, nodes.data_ptr(), times.data_ptr(), n, k, on.data_ptr(), oe.data_ptr(),
                                                                   ot.data_ptr(), st)), args.reps)
        hist = np.minimum(deg, k)
        algo = float(probe_bytes.sum() + 16 * hist.sum() + 20.0 * k * n + 16 * n)
        print(json.dumps({"graph": gname, "kernel": f"dygnn_sample_recent k={k}", "queries": n, "queries_per_s": round(n / sec),
                          "algorithmic_GBps": round(algo / sec / 1e9, 1), "frac_of_hbm_peak": round(algo / sec / HBM_PEAK, 4),
                          "ms": round(sec * 1e3, 3)}))
    L = 64
    hist_t = torch.empty(n, dtype=torch.int32, device=dev)
    end_t = torch.empty(n, dtype=torch.int64, device=dev)
    maxw = torch.zeros(1, dtype=torch.int32, device=dev)
    S = 64
    ids = torch.empty((n, S), dtype=torch.int64, device=dev)
    eids = torch.empty_like(ids)
    ts = torch.empty((n, S), dtype=torch.float32, device=dev)

    def windows():
        _capi.check(lib.dygnn_window_lengths(csr, nodes.data_ptr(), times.data_ptr(), n, L, hist_t.data_ptr(), end_t.data_ptr(), maxw.data_ptr(), st))
        _capi.check(lib.dygnn_window_fill(csr, nodes.data_ptr(), times.data_ptr(), n, L, S, hist_t.data_ptr(), end_t.data_ptr(), ids.data_ptr(),
                                          eids.data_ptr(), ts.data_ptr(), st))
    sec = timed(windows, args.reps)
    hist = np.minimum(deg, L - 1)
    algo = float(probe_bytes.sum() + 16 * hist.sum() + 20.0 * S * n + 16 * n + 12 * n)
    print(json.dumps({"graph": gname, "kernel": "dygnn_window_lengths + dygnn_window_fill (L=64)", "queries": n, "queries_per_s": round(n / sec),
                      "algorithmic_GBps": round(algo / sec / 1e9, 1), "frac_of_hbm_peak": round(algo / sec / HBM_PEAK, 4), "ms": round(sec * 1e3, 3)}))
    npair = n // 2
    a, b = ids[:npair], ids[npair:2 * npair]
    sec = timed(lambda: count_nodes_appearances(a, b, device=dev), max(1, args.reps // 2))
    algo = float(npair * (2 * S * 8 + 2 * S * 2 * 4))
    print(json.dumps({"graph": gname, "kernel": "dygnn_cooccurrence (S=64+64)", "pairs": npair, "pairs_per_s": round(npair / sec),
                      "algorithmic_GBps": round(algo / sec / 1e9, 1), "frac_of_hbm_peak": round(algo / sec / HBM_PEAK, 4), "ms": round(sec * 1e3, 3)}))
