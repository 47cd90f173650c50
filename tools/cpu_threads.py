import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from dyglib_amd import synthetic as syn
from oracle import dygformer_oracle as orc
data, nf, ef = syn.make_bipartite_graph(8227, 1000, 157474, seed=0)
params = {k: torch.from_numpy(v) for k, v in syn.make_dygformer_params(0, patch_size=2).items()}
mp = {k: torch.from_numpy(v) for k, v in syn.make_merge_layer_params(1000).items()}
adj = orc.OracleAdjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
nft, eft = torch.from_numpy(nf), torch.from_numpy(ef)
E = data.num_interactions
rs = np.random.RandomState(2); ud = np.unique(data.dst_node_ids)
def batch(i):
    sl = slice(E - 200 * (i + 1), E - 200 * i)
    return data.src_node_ids[sl], data.dst_node_ids[sl], syn.random_negative_dst(rs, ud, 200), data.node_interact_times[sl]
for th in (1, 8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    orc.link_prediction_step(params, mp, nft, eft, adj, *batch(0), 2, 64)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 4:
        orc.link_prediction_step(params, mp, nft, eft, adj, *batch(n + 1), 2, 64); n += 1
    print(th, 'threads:', round(n * 200 / (time.perf_counter() - t0), 1), 'edges/s', flush=True)
