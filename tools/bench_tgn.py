#!/usr/bin/env python3
"""Secondary benchmark (BASELINE config 5): TGN link-prediction forward on a MOOC-shaped synthetic graph
(7,047 + 97 nodes, 411,749 edges, 4 non-zero edge-feature columns), k = 10, 1 layer, batch 200, batches strictly in
chronological order from interaction 0: negative call + positive call (memory update) + MergeLayer+sigmoid per step.
TGN does not shard: "replicas only" (SURVEY.md §8e)."""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dyglib_amd import MemoryModel, MergeLayer, get_neighbor_sampler, synthetic as syn

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=100); ap.add_argument("--warmup", type=int, default=10); ap.add_argument("--cpu-steps", type=int, default=10)
ap.add_argument("--two-calls", action="store_true", help="negative call then positive call per step, as the reference issues them")
ap.add_argument("--graph", action="store_true", help="replay one captured HIP graph per step instead of issuing its ~50 launches (measured: no gain, "
                "257 k vs 262 k edges/s -- the step is bound by the GPU-side latency of ~50 dependent small kernels, not by the host)")
args = ap.parse_args()
dev, B, K = "cuda:0", 200, 10
data, nf, ef = syn.make_bipartite_graph(7047, 97, 411749, seed=0, edge_feat_kind="sparse4")
params, mparams = syn.make_tgn_params(0, nf.shape[0], num_layers=1), syn.make_merge_layer_params(1000)
sampler = get_neighbor_sampler(data, "recent", seed=1, device=dev)
model = MemoryModel(nf, ef, sampler, 100, model_name="TGN", num_layers=1, num_heads=2, dropout=0.1, device=dev)
sd = model.state_dict(); sd.update({k: torch.from_numpy(v) for k, v in params.items()}); model.load_state_dict(sd)
merge = MergeLayer(172, 172, 172, 1); merge.load_state_dict({k: torch.from_numpy(v) for k, v in mparams.items()})
model, merge = model.to(dev).eval(), merge.to(dev).eval()
rs = np.random.RandomState(2); ud = np.unique(data.dst_node_ids)
n = args.steps + args.warmup
host = [(data.src_node_ids[i * B:(i + 1) * B], data.dst_node_ids[i * B:(i + 1) * B], syn.random_negative_dst(rs, ud, B),
         data.node_interact_times[i * B:(i + 1) * B], data.edge_ids[i * B:(i + 1) * B]) for i in range(n)]
batches = [tuple(torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in b) for b in host]
def step(i):
    return step_on(*batches[i])
def step_on(s, d, ng, t, e):
    with torch.no_grad():
        if args.two_calls:       # the reference's call pattern: negative call, then positive call (evaluate_models_utils.py:85-107)
            a, b_ = model.compute_src_dst_node_temporal_embeddings(s, ng, t, edge_ids=None, edges_are_positive=False, num_neighbors=K)
            c, f = model.compute_src_dst_node_temporal_embeddings(s, d, t, edge_ids=e, edges_are_positive=True, num_neighbors=K)
            return merge.link_probabilities(c, f), merge.link_probabilities(a, b_)
        c, f, a, b_ = model.compute_step_embeddings(s, d, s, ng, t, e, num_neighbors=K)      # both calls as one (same state, written once at the end)
        p = merge.link_probabilities(torch.cat([c, a]), torch.cat([f, b_]))
        return p[:B], p[B:]
model.memory_bank.__init_memory_bank__()
for i in range(args.warmup): step(i)
run = step
if args.graph:                       # same kernels, same arguments: one hipGraph launch per step instead of ~50 kernel launches
    from dyglib_amd.graphs import GraphedStep
    graphed = GraphedStep(lambda s, d, ng, t, e: step_on(s, d, ng, t, e), batches[args.warmup])
    run = lambda i: graphed(*batches[i])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(args.steps): run(args.warmup + i)
torch.cuda.synchronize(); el = time.perf_counter() - t0
out = {"metric": "edges/sec (link-prediction fwd) TGN MOOC-shaped", "value": round(args.steps * B / el, 1), "unit": "edges/s", "n_gpus": 1,
       "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 3), "dtype": "f32", "data": "synthetic", "scaling": "replicas only", "hip_graph": args.graph,
       "config": {"workload": "TGN link-prediction forward, synthetic MOOC-shaped graph (7047+97 nodes, 411749 edges), k=10, 1 layer, batch=200, sequential batches"}}
if args.cpu_steps > 0:
    import bench                                   # the CPU-baseline leg lives in bench.py (the only non-test user of oracle/)
    out["cpu_baseline"] = bench.cpu_baseline_tgn(data, nf, ef, params, host, K, args.cpu_steps, B)
print(json.dumps(out))
