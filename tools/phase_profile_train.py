#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the fused TRAINING forward (k_dygformer_fused3<.., true>) from in-kernel s_memtime stamps.
Needs the stamps build:  python -m dyglib_amd._build --variant=stamps ; read the SHARES, not the times."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DYGNN_LIB_VARIANT", "stamps")
dev = "cuda:0"
stamps = torch.zeros((4, 8, 32), dtype=torch.int64, device=dev)
os.environ["DYGNN_STAMPS_PTR"] = hex(stamps.data_ptr())
from dyglib_amd import DyGFormer, get_neighbor_sampler, synthetic as syn  # noqa: E402

data, nf, ef = syn.make_bipartite_graph(8227, 1000, 157474, seed=0)
params = syn.make_dygformer_params(0, patch_size=2)
sampler = get_neighbor_sampler(data, "recent", seed=1, device=dev)
model = DyGFormer(nf, ef, sampler, 100, 50, patch_size=2, num_layers=2, num_heads=2, dropout=float(os.environ.get("PHASE_DROPOUT", "0.1")),
                  max_input_sequence_length=64, device=dev)
model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
model = model.to(dev).train()
E = data.num_interactions
sl = slice(E - 400, E)
src, dst, t = data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl]
for _ in range(4):
    a, b = model.compute_src_dst_node_temporal_embeddings(src, dst, t)
torch.cuda.synchronize()
st = stamps.cpu().numpy().astype(np.int64)
cats = ["windows+counts", "projection", "layernorm", "QKV (+K/V store)", "barrier after K/V", "attention (S,softmax,PV)",
        "out-projection", "FFN", "mean+output layer", "misc (param copies, taps)", "pool shuffles", "pool barrier", "proj node", "proj time", "proj edge", "proj cooc",
        "FFN: W1 MFMAs", "FFN: GELU (+dropout, stores)", "FFN: barrier+refill after W1", "FFN: W2 MFMAs", "FFN: barrier+refill after W2"]
tot = st[:, :, 31].astype(np.float64)
print(f"total ticks per wave: mean {tot.mean():.0f} (min {tot.min():.0f} max {tot.max():.0f})")
for i, nm in enumerate(cats):
    v = st[:, :, i].astype(np.float64)
    print(f"{nm:28s} {v.mean():12.0f} {100 * v.mean() / tot.mean():6.1f}%   (per-wave min {v.min():.0f} max {v.max():.0f})")
