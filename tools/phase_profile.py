#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the fused DyGFormer kernel from in-kernel s_memtime stamps.
Needs the stamps build:  python -m dyglib_amd._build --variant=stamps
Run:  DYGNN_LIB_VARIANT=stamps python tools/phase_profile.py
Never quote this build's run time (the stamps perturb it): read the SHARES."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DYGNN_LIB_VARIANT", "stamps")
from dyglib_amd import DyGFormer, get_neighbor_sampler, synthetic as syn  # noqa: E402

dev = "cuda:0"
# PHASE_WORKLOAD=lastfm: BASELINE config 4's shape (L=512, P=8: one 128-token pair per workgroup, k_dygformer_fused3<8>)
WL = {"wikipedia": (8227, 1000, 157474, 64, 2, "normal"), "lastfm": (980, 1000, 1293103, 512, 8, "zeros")}[os.environ.get("PHASE_WORKLOAD", "wikipedia")]
data, nf, ef = syn.make_bipartite_graph(WL[0], WL[1], WL[2], seed=0, edge_feat_kind=WL[5])
params = syn.make_dygformer_params(0, patch_size=WL[4])
sampler = get_neighbor_sampler(data, "recent", seed=1, device=dev)
model = DyGFormer(nf, ef, sampler, 100, 50, patch_size=WL[4], num_layers=2, num_heads=2, dropout=0.1,
                  max_input_sequence_length=WL[3], device=dev)
model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
model = model.to(dev).eval()
model.impl = int(os.environ.get("PHASE_IMPL", "3"))
E = data.num_interactions
NG = int(os.environ.get("PHASE_GROUPS", "16"))     # groups of 200 pairs per launch (bench default: 16)
sl = slice(E - 200 * NG, E)
src, dst, t = data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl]
with torch.no_grad():
    for _ in range(3):
        model.compute_src_dst_node_temporal_embeddings(src, dst, t, _group_size=200)
    taps = {"want_phase_cycles": True}
    # the LAST four workgroups of the grid are stamped: steady state (their neighbours are in unrelated phases)
    model.compute_src_dst_node_temporal_embeddings(src, dst, t, _taps=taps, _group_size=200)
torch.cuda.synchronize()
st = taps["phase_cycles"].cpu().numpy().astype(np.int64)      # [4 wg][8 waves][32]
if model.impl == 3:
    cats = ["windows+counts", "projection", "layernorm", "QKV (+K/V store)", "barrier after K/V", "attention (S,softmax,PV)",
            "out-projection", "FFN", "mean+output layer", "misc (param copies, taps)", "pool shuffles", "pool barrier", "proj node", "proj time", "proj edge", "proj cooc",
            "FFN: W1 MFMAs", "FFN: GELU", "FFN: barrier+refill after W1", "FFN: W2 MFMAs", "FFN: barrier+refill after W2"]
    tot = st[:, :, 31].astype(np.float64)
    print(f"total ticks per wave: mean {tot.mean():.0f} (min {tot.min():.0f} max {tot.max():.0f})")
    for i, nm in enumerate(cats):
        v = st[:, :, i].astype(np.float64)
        print(f"{nm:28s} {v.mean():12.0f} {100 * v.mean() / tot.mean():6.1f}%   (per-wave min {v.min():.0f} max {v.max():.0f})")
    print("per wave (mean over the 4 stamped workgroups), waves 0..7:")
    for i in (3, 6, 16, 17, 18, 19, 20):
        print(f"{cats[i]:28s} " + " ".join(f"{st[:, w, i].mean():9.0f}" for w in range(8)))
    sys.exit(0)
NL = 2
names = ["zero+windows+counts", "projection", "barrier"]
for l in range(NL):
    names += [f"L{l} LN0", f"L{l} QKV", f"L{l} barrier(qkv)", f"L{l} attention+outproj", f"L{l} exchange1",
              f"L{l} LN1", f"L{l} FFN", f"L{l} exchange2+tap"]
names += ["pool+output"]
n = len(names) + 1
d = np.diff(st[:, :, :n], axis=2).astype(np.float64)          # [4][8][n-1]
tot = (st[:, :, n - 1] - st[:, :, 0]).astype(np.float64)
print(f"total cycles per wave: mean {tot.mean():.0f} (min {tot.min():.0f} max {tot.max():.0f})")
print(f"{'phase':28s} {'hf=0 waves':>12s} {'hf=1 waves':>12s} {'share':>7s}")
for i, nm in enumerate(names):
    a, b = d[:, :4, i].mean(), d[:, 4:, i].mean()
    print(f"{nm:28s} {a:12.0f} {b:12.0f} {100 * d[:, :, i].mean() / tot.mean():6.1f}%")

sub = st[:, :, 24:30].astype(np.float64)
for i, nm in enumerate(["QKV k-loop (2 layers)", "QKV epilogue", "FFN A stages", "FFN A barrier", "FFN B stages+barrier"]):
    print(f"  sub: {nm:26s} hf0 {sub[:, :4, i].mean():10.0f}  hf1 {sub[:, 4:, i].mean():10.0f}")
