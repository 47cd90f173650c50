#!/usr/bin/env python3
"""Headline benchmark: positive edges / second through the DyGFormer link-prediction forward step
(BASELINE.json metric; body of reference evaluate_models_utils.py:126-141) on the Wikipedia-shaped
synthetic workload of SURVEY.md §8(d): L=64, P=2, batch 200, 2 layers, 2 heads, C=50.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One step = one 200-edge batch: hot-path call on (src,dst,t) + hot-path call on (src,neg_dst,t) +
sigmoid(MergeLayer) on both + the per-step metric (AUC numerator) reduced over RCCL when N>1.
Inputs (graph, feature tables, weights, every batch's id/time arrays) are resident in HBM before the
timed region.  N>1: one process per GPU, graph/tables/weights replicated, whole batches dealt
round-robin (rank r takes batches r, r+N, ...: weak scaling, K steps per rank), no data-path
collective.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from dyglib_amd import distributed as D  # noqa: E402
from dyglib_amd import synthetic as syn  # noqa: E402

# algorithmic work of one (src,dst,t) pair at S_src = S_dst = 64 (every batch of this
# workload), multiply-add = 2 flops; and the fp32 MFMA peak of MI355X_MICROARCH.md.
FLOP_PER_PAIR = 137.2e6
PEAK_F32_MFMA_TFLOPS = 157.3

WORKLOADS = {
    # name: users, items, edges, L, P, batch
    "wikipedia": dict(users=8227, items=1000, edges=157474, L=64, P=2, batch=200),
    "lastfm": dict(users=980, items=1000, edges=1293103, L=512, P=8, batch=200),  # BASELINE config 4 shape (128 tokens per pair)
    "tiny": dict(users=300, items=50, edges=12000, L=64, P=2, batch=200),        # CI / smoke sizes
}


def flops_per_pair(S_s: int, S_d: int, P: int, Fn=172, Ft=100, C=50, layers=2) -> float:
    """SURVEY.md §8(d) formula."""
    S, D = S_s + S_d, 4 * C
    T = S // P
    return (2 * Ft * S + S * 2 * (2 * C + 2 * C * C) + T * 2 * C * P * (2 * Fn + Ft + C)
            + layers * (T * (2 * D * 3 * D + 2 * D * D + 4 * D * 4 * D) + 4 * T * T * D) + 2 * 2 * D * Fn)


# ---- CPU-baseline legs of the secondary benchmarks (tools/bench_tgat.py, bench_tgn.py, bench_train.py).  They live here because
# bench.py's cpu_baseline leg is the only non-test code allowed to execute anything under oracle/.
def cpu_baseline_tgat(data, nf, ef, params, host_batches, k: int, steps: int, B: int) -> dict:
    from oracle import dygformer_oracle as orc, tgat_oracle as torc
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    adj = orc.OracleAdjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    nft, eft = torch.from_numpy(nf), torch.from_numpy(ef)
    tp = {kk: torch.from_numpy(v) for kk, v in params.items()}
    t0 = time.perf_counter()
    for i in range(steps):
        s, d, n, t = host_batches[i]
        torc.tgat_forward(tp, nft, eft, adj, s, d, t, 2, k, 2)
        torc.tgat_forward(tp, nft, eft, adj, s, n, t, 2, k, 2)
    cel = time.perf_counter() - t0
    return {"value": round(steps * B / cel, 1), "unit": "edges/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} of the same steps ({cel:.1f} s), oracle/tgat_oracle.py"}


def cpu_baseline_tgn(data, nf, ef, params, host_batches, k: int, steps: int, B: int) -> dict:
    from oracle import dygformer_oracle as orc, tgn_oracle as tn
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    adj = orc.OracleAdjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    nft, eft = torch.from_numpy(nf), torch.from_numpy(ef)
    tp = {kk: torch.from_numpy(v) for kk, v in params.items()}
    st = tn.TgnState(nf.shape[0], 172)
    t0 = time.perf_counter()
    for i in range(steps):
        s, d, ng, t, e = host_batches[i]
        tn.tgn_forward(tp, nft, eft, adj, st, s, ng, t, None, False, 1, k, 2)
        tn.tgn_forward(tp, nft, eft, adj, st, s, d, t, e, True, 1, k, 2)
    cel = time.perf_counter() - t0
    return {"value": round(steps * B / cel, 1), "unit": "edges/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"the first {steps} steps ({cel:.1f} s), oracle/tgn_oracle.py"}


def cpu_baseline_train(data, nf, ef, params, mparams, batch_fn, P: int, L: int, steps: int, B: int) -> dict:
    """the training step (pos + neg forward, BCE, backward, Adam) through torch autograd over the oracle"""
    from oracle import dygformer_oracle as orc
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cp = {kk: torch.from_numpy(v.copy()).requires_grad_(True) for kk, v in params.items()}
    cm = {kk: torch.from_numpy(v.copy()).requires_grad_(True) for kk, v in mparams.items()}
    adj = orc.OracleAdjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    copt = torch.optim.Adam(list(cp.values()) + list(cm.values()), lr=1e-4)

    def cpu_step(i):
        src, dst, neg, t = batch_fn(i)
        ps, pd = orc.dygformer_forward(cp, nf, ef, adj, src, dst, t, P, L)
        ns, nd = orc.dygformer_forward(cp, nf, ef, adj, src, neg, t, P, L)
        pos, ng = orc.merge_layer(cm, ps, pd).squeeze(-1).sigmoid(), orc.merge_layer(cm, ns, nd).squeeze(-1).sigmoid()
        loss = torch.nn.functional.binary_cross_entropy(torch.cat([pos, ng]), torch.cat([torch.ones_like(pos), torch.zeros_like(ng)]))
        copt.zero_grad()
        loss.backward()
        copt.step()
    cpu_step(0)
    c0 = time.perf_counter()
    for i in range(steps):
        cpu_step(1 + i)
    csec = (time.perf_counter() - c0) / steps
    return {"value": round(B / csec, 1), "unit": "edges/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} of the same steps ({csec * steps:.1f} s) through oracle autograd"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="wikipedia", choices=list(WORKLOADS))
    ap.add_argument("--impl", type=int, default=0, help="0 auto, 1 generic kernels, 2 fused kernel (wave-pair), 3 fused kernel (token-owner)")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the launches are issued on (round-robin)")
    ap.add_argument("--fuse-steps", type=int, default=32,
                    help="steps per launch: the positive and negative calls of F consecutive steps (2F independently "
                         "padded groups of `batch` pairs) form ONE grid, so 256 CUs stay busy instead of 200")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="wall budget of the CPU-baseline sample (0 = skip)")
    return ap.parse_args()


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for N>1 launch with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from dyglib_amd import DyGFormer, MergeLayer, get_neighbor_sampler
    wl = WORKLOADS[args.workload]
    B, L, P = wl["batch"], wl["L"], wl["P"]
    data, node_feat, edge_feat = syn.make_bipartite_graph(wl["users"], wl["items"], wl["edges"], seed=0)
    params = syn.make_dygformer_params(0, patch_size=P)
    mparams = syn.make_merge_layer_params(1000)
    sampler = get_neighbor_sampler(data, "recent", seed=1, device=dev)              # full graph, as in evaluation
    model = DyGFormer(node_feat, edge_feat, sampler, time_feat_dim=100, channel_embedding_dim=50, patch_size=P,
                      num_layers=2, num_heads=2, dropout=0.1, max_input_sequence_length=L, device=dev)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    merge = MergeLayer(172, 172, 172, 1)
    merge.load_state_dict({k: torch.from_numpy(v) for k, v in mparams.items()})
    model, merge = model.to(dev).eval(), merge.to(dev).eval()
    model.impl = args.impl

    # evaluation span = last 30 % of the interactions (val + test), full batches only
    E = data.num_interactions
    first = int(E * 0.70)
    n_batches = (E - first) // B
    neg_rs = np.random.RandomState(2)
    uniq_dst = np.unique(data.dst_node_ids)
    batches = []
    for i in range(n_batches):
        sl = slice(first + i * B, first + (i + 1) * B)
        neg = syn.random_negative_dst(neg_rs, uniq_dst, B)
        batches.append((data.src_node_ids[sl], data.dst_node_ids[sl], neg, data.node_interact_times[sl]))
    # device-resident inputs [n_batches, B]; a launch takes F consecutive steps of this rank
    src_all = torch.from_numpy(np.stack([b[0] for b in batches])).to(dev)
    dst_all = torch.from_numpy(np.stack([b[1] for b in batches])).to(dev)
    neg_all = torch.from_numpy(np.stack([b[2] for b in batches])).to(dev)
    t_all = torch.from_numpy(np.stack([b[3] for b in batches])).to(dev)
    F = max(1, args.fuse_steps)
    order = torch.tensor([(k * world + rank) % n_batches for k in range(args.warmup + args.steps)], device=dev)
    streams = [torch.cuda.Stream(dev) for _ in range(args.streams)] if args.streams > 1 else [torch.cuda.current_stream(dev)]
    # per-stream accumulators [sum AUC, sum mean-prob gap, steps] (no cross-stream read-modify-write)
    metric_accs = [torch.zeros(3, dtype=torch.float64, device=dev) for _ in streams]
    from dyglib_amd import link_prediction_metrics_device
    labels_full = torch.cat([torch.ones(F, B), torch.zeros(F, B)], dim=1).to(dev)          # evaluate_models_utils.py:143

    def launch(first_step: int, nsteps: int, li: int, ev=None):
        """steps first_step .. first_step+nsteps-1 of this rank as ONE hot-path launch (2*nsteps groups)."""
        idx = order[first_step:first_step + nsteps]
        st = streams[li % len(streams)]
        with torch.cuda.stream(st), torch.no_grad():
            src = src_all[idx]
            srcs = torch.cat([src, src])                       # negative sources = batch sources (evaluate_models_utils.py:62-63)
            dsts = torch.cat([dst_all[idx], neg_all[idx]])
            ts = torch.cat([t_all[idx], t_all[idx]])
            if ev is not None:
                ev[0].record(st)
            s, d = model.compute_src_dst_node_temporal_embeddings_many(srcs, dsts, ts)      # [2n, B, 172]
            if ev is not None:
                ev[1].record(st)
            prob = merge.link_probabilities(s.reshape(-1, s.shape[-1]), d.reshape(-1, d.shape[-1])).reshape(2, nsteps, B)
            pos, negp = prob[0], prob[1]
            # per-step ROC AUC on the device (dygnn_link_metrics: evaluate_models_utils.py:139-150 without the host round trip), reduced over RCCL
            predicts = torch.cat([pos, negp], dim=1)
            labels = labels_full[:nsteps]
            _, auc, _, _ = link_prediction_metrics_device(predicts, labels)
            m = torch.stack([auc.sum(), (pos.mean(dim=1) - negp.mean(dim=1)).double().sum(),
                             torch.full((), float(nsteps), dtype=torch.float64, device=dev)])
            D.reduce_metric_sums(m)                              # RCCL all-reduce of 3 float64 when N > 1
            metric_accs[li % len(streams)].add_(m)

    def run_steps(first: int, count: int, evs=None):
        li, done = 0, 0
        while done < count:
            n = min(F, count - done)
            launch(first + done, n, li, None if evs is None else evs[li])
            done += n
            li += 1
        return li

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    run_steps(0, args.warmup)
    sync_all()
    [a.zero_() for a in metric_accs]
    n_launch = (args.steps + F - 1) // F
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_launch)]
    t0 = time.perf_counter()
    run_steps(args.warmup, args.steps, events)
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    total_edges = args.steps * B * world
    value = total_edges / elapsed
    # dominant kernel = the fused forward: one launch per F steps.  Duration from the HIP events recorded on the launch
    # stream inside the timed region (they bracket the window-search + fused-forward launches of one call).
    full = [i for i in range(n_launch) if min(F, args.steps - i * F) == F] or list(range(n_launch))
    launch_ms = float(np.mean([events[i][0].elapsed_time(events[i][1]) for i in full]))
    steps_per_launch = F if full != list(range(n_launch)) or args.steps >= F else args.steps
    fpp = flops_per_pair(L, L, P)
    flop_per_launch = fpp * B * 2 * steps_per_launch
    achieved_tflops = flop_per_launch / (launch_ms * 1e-3) / 1e12
    # HBM traffic of the fused kernel from the committed PMC passes (profiles/*_traffic.json, measured with
    # tools/pmc_profile.sh on this command); null when the launch shape differs from the profiled one
    traffic = None
    try:
        import glob
        tf = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))[-1]
        tj = json.load(open(tf))
        if args.workload == "wikipedia" and args.impl in (0, 2, 3):
            traffic = tj["hbm_bytes_per_pair"] * 2 * steps_per_launch * B
    except Exception:
        traffic = None
    acc = sum(a.cpu().numpy() for a in metric_accs)

    out = {
        "metric": "edges/sec (link-prediction fwd) DyGFormer " + {"wikipedia": "Wikipedia", "lastfm": "LastFM-shaped (config 4)", "tiny": "tiny"}[args.workload],
        "value": round(value, 1), "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"DyGFormer link-prediction forward, synthetic {args.workload}-shaped graph "
                               f"({wl['users']}+{wl['items']} nodes, {wl['edges']} edges), L={L}, P={P}, batch={B}, "
                               f"2 layers, 2 heads, C=50; pos+neg calls + MergeLayer+sigmoid per step",
                   "batch": B, "max_input_sequence_length": L, "patch_size": P,
                   "parallelism": f"{world} x edge-batch shard, graph+weights replicated, metric all-reduce over RCCL",
                   "impl": {0: "auto", 1: "generic", 2: "fused", 3: "fused3"}[args.impl], "streams": len(streams),
                   "steps_per_launch": F},
        "roofline": {"bound": "mfma", "achieved": round(achieved_tflops, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved_tflops / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                     "kernel": {1: "generic multi-kernel path", 2: "k_dygformer_fused"}.get(args.impl, "k_dygformer_fused3<%d>" % (4 if 2 * ((L + P - 1) // P) <= 64 else 8)) + " (+ the 3 tiny window-search launches in front of it)",
                     "flop_per_launch": flop_per_launch, "ms_per_launch": round(launch_ms, 4),
                     "pairs_per_launch": 2 * steps_per_launch * B},
        "mean_auc": round(float(acc[0] / max(acc[2], 1)), 4),
    }

    if rank == 0 and world == 1:
        out["stages"] = sampler_stage(sampler, data, dev)           # stage-level number for the neighbour lookup
        out["stages"].update(metrics_stage(dev))
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        out["cpu_baseline"] = cpu_baseline(params, mparams, node_feat, edge_feat, data, batches, L, P, args.cpu_seconds)
        out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def sampler_stage(sampler, data, dev, n_queries: int = 400_000, k: int = 20, reps: int = 5) -> dict:
    """get_historical_neighbors ('recent', k = 20) on random (endpoint, time) queries of the evaluation span: queries/s and the
    algorithmic-byte rate of DESIGN.md §4.1 against the 8 TB/s HBM peak (the reference: 149 k queries/s on the CPU)."""
    E = data.num_interactions
    rs = np.random.RandomState(1)
    idx = rs.randint(int(0.7 * E), E, size=n_queries // 2)
    nodes_h = np.concatenate([data.src_node_ids[idx], data.dst_node_ids[idx]])
    times_h = np.concatenate([data.node_interact_times[idx], data.node_interact_times[idx]])
    nodes, times = torch.from_numpy(nodes_h).to(dev), torch.from_numpy(times_h).to(dev)
    sampler.get_historical_neighbors_device(nodes, times, k)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        sampler.get_historical_neighbors_device(nodes, times, k)
    e1.record()
    torch.cuda.synchronize(dev)
    sec = e0.elapsed_time(e1) * 1e-3 / reps
    deg = (sampler.csr.indptr[nodes_h + 1] - sampler.csr.indptr[nodes_h]).astype(np.float64)
    probes = np.maximum(1, np.ceil(np.log(deg + 1) / np.log(64)))
    algo = float((8 * 64 * probes).sum() + 16 * np.minimum(deg, k).sum() + 20.0 * k * len(nodes_h) + 16 * len(nodes_h))
    return {"sampler_recent_k20_queries_per_s": round(len(nodes_h) / sec), "sampler_algorithmic_GBps": round(algo / sec / 1e9, 1),
            "sampler_frac_of_hbm_peak": round(algo / sec / 8.0e12, 4), "queries": len(nodes_h)}


def metrics_stage(dev, groups: int = 237, n: int = 400, reps: int = 10) -> dict:
    """AP + ROC AUC + BCELoss of `groups` evaluation batches (200 positive + 200 negative scores each) in one launch
    (dygnn_link_metrics): batches/s, outside the headline metric (SURVEY §8(d) excludes the sklearn metrics)."""
    from dyglib_amd import link_prediction_metrics_device
    g = torch.Generator(device="cpu").manual_seed(0)
    y = torch.cat([torch.ones(groups, n // 2), torch.zeros(groups, n // 2)], dim=1).to(dev)
    p = torch.sigmoid(torch.randn(groups, n, generator=g)).to(dev)
    link_prediction_metrics_device(p, y)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        link_prediction_metrics_device(p, y)
    e1.record()
    torch.cuda.synchronize(dev)
    return {"metrics_batches_per_s": round(groups * reps / (e0.elapsed_time(e1) * 1e-3))}


def cpu_baseline(params, mparams, node_feat, edge_feat, data, batches, L, P, budget_s):
    """The CPU oracle (restatement of the reference path, kind 'port') timed on this host on a bounded
    sample of the SAME workload: as many of the same 200-edge steps as fit in ~budget_s seconds."""
    from oracle import dygformer_oracle as orc
    adj = orc.OracleAdjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    nf, ef = torch.from_numpy(node_feat), torch.from_numpy(edge_feat)
    tp = {k: torch.from_numpy(v) for k, v in params.items()}
    mp = {k: torch.from_numpy(v) for k, v in mparams.items()}
    # torch's default (every logical CPU of the host, 256 on the GPU box) oversubscribes these small ops and is 6x
    # slower than 16 threads — the box's CPU share for one GPU and the measured optimum (tools/cpu_threads.py).
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    src, dst, neg, t = batches[0]
    orc.link_prediction_step(tp, mp, nf, ef, adj, src, dst, neg, t, P, L)        # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        src, dst, neg, t = batches[(n + 1) % len(batches)]
        orc.link_prediction_step(tp, mp, nf, ef, adj, src, dst, neg, t, P, L)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 64:
            break
    return {"value": round(n * len(src) / el, 1), "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": f"{n} of the same 200-edge steps ({el:.1f} s) after 1 warm-up step; oracle/dygformer_oracle.py "
                      f"(numpy sampling + PyTorch-CPU fp32 dense ops, torch threads={cores})"}


if __name__ == "__main__":
    main()
