#!/usr/bin/env python3
"""A/B of builds of the fused DyGFormer kernel in ONE process (interleaved rounds, cdna_hip_programming.md §5.4 rule 24):
    python tools/ab_fused3.py [variant ...]          # names of dyglib_amd/_build.py VARIANTS; "" or "default" = the shipped build
Build the variants on the CPU box first (python -m dyglib_amd._build --variant=NAME): the .so files travel with gpurun.
Each arm runs the bench.py launch shape (32 steps = 64 groups of 200 pairs = 6,400 workgroups on the Wikipedia-shaped workload);
prints per-arm median / min ms per launch, edges/s, the fraction of the fp32-MFMA peak, and max |out - out(arm 0)|."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dyglib_amd import _capi  # noqa: E402

names = [("" if a == "default" else a) for a in sys.argv[1:]] or ["", "f3base"]
rounds = int(os.environ.get("AB_ROUNDS", "9"))
F = int(os.environ.get("AB_STEPS", "32"))
workload = os.environ.get("AB_WORKLOAD", "wikipedia")
PAIRED = os.environ.get("AB_PAIRED", "1") != "0"      # positive and negative pair of an edge in one workgroup (f4)
dev = torch.device("cuda", 0)
wk = bench.DygformerWorkload(workload, dev)
arms = []
for nm in names:
    from dyglib_amd import DyGFormer
    m = DyGFormer(wk.node_feat, wk.edge_feat, wk.sampler, time_feat_dim=100, channel_embedding_dim=50, patch_size=wk.P, num_layers=2, num_heads=2,
                  dropout=0.1, max_input_sequence_length=wk.L, device=dev)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in wk.params.items()})
    m = m.to(dev).eval()
    if nm:
        m._lib = _capi.load_variant(nm)
    arms.append(m)
idx = torch.arange(F, device=dev) % wk.n_batches
src = wk.src_all[idx]
srcs, dsts, ts = torch.cat([src, src]), torch.cat([wk.dst_all[idx], wk.neg_all[idx]]), torch.cat([wk.t_all[idx], wk.t_all[idx]])
outs, times = [], [[] for _ in arms]
with torch.no_grad():
    for m in arms:
        outs.append(m.compute_src_dst_node_temporal_embeddings_many(srcs, dsts, ts, pos_neg_halves=PAIRED))
    torch.cuda.synchronize()
    for r in range(rounds):
        for i, m in enumerate(arms):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(2):
                m.compute_src_dst_node_temporal_embeddings_many(srcs, dsts, ts, pos_neg_halves=PAIRED)
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / 2)
S = ((wk.L + wk.P - 1) // wk.P) * wk.P
flop = bench.flops_per_pair(S, S, wk.P) * 2 * F * wk.B
for i, nm in enumerate(names):
    t = np.array(times[i])
    diff = max(float((outs[i][0] - outs[0][0]).abs().max()), float((outs[i][1] - outs[0][1]).abs().max()))
    print(json.dumps({"arm": nm or "default", "ms_median": round(float(np.median(t)), 4), "ms_min": round(float(t.min()), 4),
                      "edges_per_s_median": round(F * wk.B / np.median(t) * 1e3), "frac_of_peak_median": round(flop / (np.median(t) * 1e-3) / 157.3e12, 4),
                      "vs_arm0": round(float(np.median(times[0]) / np.median(t)), 4), "max_abs_diff_vs_arm0": diff}))
