#!/usr/bin/env python3
"""Where the time of k_tgat_post goes: s_memtime stamps of thread 0 of every workgroup at the stage boundaries (DYGNN_CHAIN_STAMPS = the
address of a device buffer of 16 uint64 per workgroup) over steps of bench.py's TGN workload.  Shares of the kernel's cycles (s_memtime
runs at the shader clock here: the sum is ~63 k ticks for a 30-us kernel)."""
import os, sys
os.environ.setdefault("DYGNN_LIB_VARIANT", "stamps")      # the hook is compiled only into the stamps build (python -m dyglib_amd._build --variant=stamps)
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dyglib_amd import MemoryModel, get_neighbor_sampler, synthetic as syn

dev = "cuda:0"
B, K = 200, 10
data, nf, ef = syn.make_bipartite_graph(7047, 97, 411749, seed=0, edge_feat_kind="sparse4")
sampler = get_neighbor_sampler(data, "recent", seed=1, device=dev)
model = MemoryModel(nf, ef, sampler, 100, model_name="TGN", num_layers=1, num_heads=2, dropout=0.1, device=dev)
sd = model.state_dict()
sd.update({k: torch.from_numpy(v) for k, v in syn.make_tgn_params(0, nf.shape[0], num_layers=1).items()})
model.load_state_dict(sd)
model = model.to(dev).eval()
rs, ud = np.random.RandomState(2), np.unique(data.dst_node_ids)
stamps = torch.zeros(1024 * 16, dtype=torch.int64, device=dev)
NAMES = ["z rows + parameters -> LDS", "W_v z", "residual rows -> LDS", "residual_fc", "LayerNorm", "fc1", "fc2 + store"]
acc = np.zeros(len(NAMES)); tot = 0.0; n = 0
bench._prime_gpu(dev)
model.memory_bank.__init_memory_bank__()
for i in range(80):
    s_, d_, t_, e_ = (data.src_node_ids[i * B:(i + 1) * B], data.dst_node_ids[i * B:(i + 1) * B], data.node_interact_times[i * B:(i + 1) * B], data.edge_ids[i * B:(i + 1) * B])
    n_ = syn.random_negative_dst(rs, ud, B)
    j = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (np.concatenate([s_, s_]), np.concatenate([d_, n_]), np.concatenate([t_, t_]), e_)]
    if i >= 60:
        os.environ["DYGNN_CHAIN_STAMPS"] = hex(stamps.data_ptr())
    with torch.no_grad():
        model.compute_step_embeddings_joint(j[0], j[1], j[2], j[3], B, num_neighbors=K)
    if i >= 60:
        torch.cuda.synchronize()
        st = stamps.cpu().numpy().reshape(-1, 16)[:200, :8].astype(np.float64)       # 800 roots / 4 rows = 200 workgroups
        d = np.diff(st, axis=1)                                                       # ticks
        acc += np.median(d, axis=0); tot += (st[:, 7].max() - st[:, 0].min()); n += 1
print(f"k_tgat_post<1>: s_memtime ticks per stage, median over the 200 workgroups of a step, mean of {n} steps; first start -> last end of a step: {tot / n:.0f} ticks")
for nm, v in zip(NAMES, acc / n):
    print(f"  {nm:32s} {v:8.0f}  {100 * v / (acc.sum() / n):5.1f} %")
print(f"  {'sum':32s} {acc.sum() / n:8.0f}")
