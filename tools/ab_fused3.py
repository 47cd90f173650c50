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
# AB_COLD=1 (default): every timed call takes the NEXT F batches of the evaluation span, as bench.py does — newer interactions bring feature
# rows no earlier launch has touched; AB_COLD=0: the same F batches every time (rows hot in L2 / Infinity Cache: reads ~4 % faster)
COLD = os.environ.get("AB_COLD", "1") != "0"
def inputs(call):
    first = (call * F) % max(1, wk.n_batches - F) if COLD else 0
    idx = (torch.arange(F, device=dev) + first) % wk.n_batches
    src = wk.src_all[idx]
    return torch.cat([src, src]), torch.cat([wk.dst_all[idx], wk.neg_all[idx]]), torch.cat([wk.t_all[idx], wk.t_all[idx]])
outs, times = [], [[] for _ in arms]
with torch.no_grad():
    for m in arms:
        outs.append(m.compute_src_dst_node_temporal_embeddings_many(*inputs(0), pos_neg_halves=PAIRED))
    torch.cuda.synchronize()
    call = 1
    for r in range(rounds):
        base = call
        for i, m in enumerate(arms):
            call = base                                   # every arm of a round sees the same inputs
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ins = [inputs(call), inputs(call + 1)]
            torch.cuda.synchronize()
            e0.record()
            for x in ins:
                m.compute_src_dst_node_temporal_embeddings_many(*x, pos_neg_halves=PAIRED)
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / 2)
            call += 2
S = ((wk.L + wk.P - 1) // wk.P) * wk.P
flop = bench.flops_per_pair(S, S, wk.P) * 2 * F * wk.B
for i, nm in enumerate(names):
    t = np.array(times[i])
    diff = max(float((outs[i][0] - outs[0][0]).abs().max()), float((outs[i][1] - outs[0][1]).abs().max()))
    print(json.dumps({"arm": nm or "default", "ms_median": round(float(np.median(t)), 4), "ms_min": round(float(t.min()), 4),
                      "edges_per_s_median": round(F * wk.B / np.median(t) * 1e3), "frac_of_peak_median": round(flop / (np.median(t) * 1e-3) / 157.3e12, 4),
                      "vs_arm0": round(float(np.median(times[0]) / np.median(t)), 4), "max_abs_diff_vs_arm0": diff}))
