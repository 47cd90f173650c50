#!/usr/bin/env python3
"""Training-step throughput of the DyGFormer path: train_link_prediction.py:229-257 in miniature on the
Wikipedia-shaped workload — positive + negative call in train mode (dropout 0.1), MergeLayer, BCE, backward, Adam step —
with the CPU restatement (oracle autograd, same step) timed beside it on a bounded sample.  One JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dyglib_amd import DyGFormer, MergeLayer, get_neighbor_sampler, synthetic as syn  # noqa: E402
import bench  # noqa: E402  (its cpu_baseline leg is the only non-test user of oracle/)

dev = "cuda:0"
B, L, P = 200, 64, 2
steps, warmup = int(os.environ.get("STEPS", "20")), 3
data, nf, ef = syn.make_bipartite_graph(8227, 1000, 157474, seed=0)
params = syn.make_dygformer_params(0, patch_size=P)
mparams = syn.make_merge_layer_params(1000)
sampler = get_neighbor_sampler(data, "recent", seed=1, device=dev)
model = DyGFormer(nf, ef, sampler, 100, 50, patch_size=P, num_layers=2, num_heads=2, dropout=0.1, max_input_sequence_length=L, device=dev)
model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
merge = MergeLayer(172, 172, 172, 1)
merge.load_state_dict({k: torch.from_numpy(v) for k, v in mparams.items()})
model, merge = model.to(dev).train(), merge.to(dev).train()
opt = torch.optim.Adam(list(model.parameters()) + list(merge.parameters()), lr=1e-4)
E = data.num_interactions
first = int(0.7 * E)
rs = np.random.RandomState(2)
uniq_dst = np.unique(data.dst_node_ids)


def batch(i):
    sl = slice(first + i * B, first + (i + 1) * B)
    return data.src_node_ids[sl], data.dst_node_ids[sl], syn.random_negative_dst(rs, uniq_dst, B), data.node_interact_times[sl]


def step(i):
    src, dst, neg, t = batch(i)
    if os.environ.get("SEPARATE_CALLS", "0") == "1":      # the reference's call pattern (train_link_prediction.py:229-239), two dense passes
        ps, pd = model.compute_src_dst_node_temporal_embeddings(src, dst, t)
        ns, nd = model.compute_src_dst_node_temporal_embeddings(src, neg, t)
    else:                                                   # both calls as one set: one dense pass when they pad to the same lengths
        s2, d2 = model.compute_src_dst_node_temporal_embeddings_many(np.stack([src, src]), np.stack([dst, neg]), np.stack([t, t]))
        ps, pd, ns, nd = s2[0], d2[0], s2[1], d2[1]
    pos, ng = merge(ps, pd).squeeze(-1).sigmoid(), merge(ns, nd).squeeze(-1).sigmoid()
    loss = torch.nn.functional.binary_cross_entropy(torch.cat([pos, ng]), torch.cat([torch.ones_like(pos), torch.zeros_like(ng)]))
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss


for i in range(warmup):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    loss = step(warmup + i)
torch.cuda.synchronize()
sec = (time.perf_counter() - t0) / steps

# CPU baseline: the same step through the oracle's autograd (torch CPU, 16 threads), 2 steps
cpu = bench.cpu_baseline_train(data, nf, ef, params, mparams, batch, P, L, 2, B) if os.environ.get("CPU_STEPS", "2") != "0" else None
print(json.dumps({"metric": "edges/sec (link-prediction TRAIN step: fwd pos+neg, bwd, Adam) DyGFormer Wikipedia-shaped", "value": round(B / sec, 1),
                  "unit": "edges/s", "ms_per_step": round(sec * 1e3, 3), "steps": steps, "dtype": "f32", "dropout": 0.1, "final_loss": round(float(loss.detach()), 4),
                  "cpu_baseline": cpu}))
