#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the fused BACKWARD kernels (k_ffn_bwd, k_attn_bwd) from in-kernel s_memtime stamps (stamps build).
The last launch of each kernel in a step (layer 0) is what remains in the buffers.  Read the SHARES, not the times."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DYGNN_LIB_VARIANT", "stamps")
dev = "cuda:0"
sf = torch.zeros((4, 8, 32), dtype=torch.int64, device=dev)
sa = torch.zeros((4, 8, 32), dtype=torch.int64, device=dev)
os.environ["DYGNN_STAMPS_FFN"] = hex(sf.data_ptr())
os.environ["DYGNN_STAMPS_ATTN"] = hex(sa.data_ptr())
from dyglib_amd import DyGFormer, get_neighbor_sampler, synthetic as syn  # noqa: E402

data, nf, ef = syn.make_bipartite_graph(8227, 1000, 157474, seed=0)
params = syn.make_dygformer_params(0, patch_size=2)
sampler = get_neighbor_sampler(data, "recent", seed=1, device=dev)
model = DyGFormer(nf, ef, sampler, 100, 50, patch_size=2, num_layers=2, num_heads=2, dropout=0.1, max_input_sequence_length=64, device=dev)
model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
model = model.to(dev).train()
E = data.num_interactions
sl = slice(E - 400, E)
src, dst, t = data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl]
for _ in range(4):
    a, b = model.compute_src_dst_node_temporal_embeddings(src, dst, t)
    (a.sum() + b.sum()).backward()
torch.cuda.synchronize()
for name, st, cats in (("k_ffn_bwd", sf, ["prologue (dX, mask, dF2 rows, ring fill)", "W2^T block", "mask + gelu' + dhpre rows", "barrier + refill", "W1^T block", "barrier + refill", "LayerNorm backward", "dgamma / dbeta"]),
                       ("k_attn_bwd", sa, ["prologue (dAo rows, ring fill)", "Wo^T -> dOa", "K, V, Q loads + barrier", "phase A (queries)", "barrier + Q / dOa exchange", "Wq^T", "phase B: dPd, dS, dV", "Wv^T", "dK", "Wk^T", "LayerNorm backward"])):
    v = st.cpu().numpy().astype(np.float64)
    tot = v[:, :, 31]
    print(f"{name}: total ticks per wave mean {tot.mean():.0f} (min {tot.min():.0f} max {tot.max():.0f})")
    for i, nm in enumerate(cats):
        print(f"  {nm:44s} {v[:, :, i].mean():10.0f} {100 * v[:, :, i].mean() / tot.mean():6.1f}%   (per-wave min {v[:, :, i].min():.0f} max {v[:, :, i].max():.0f})")
