#!/usr/bin/env python3
"""Headline benchmark: positive edges / second through the DyGFormer link-prediction forward step
(BASELINE.json metric; body of reference evaluate_models_utils.py:126-141) on the Wikipedia-shaped
synthetic workload of SURVEY.md §8(d): L=64, P=2, batch 200, 2 layers, 2 heads, C=50.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One step = one 200-edge batch: hot-path call on (src,dst,t) + hot-path call on (src,neg_dst,t) +
sigmoid(MergeLayer) on both + the per-step ROC AUC on the device; the per-rank metric sums are all-reduced over RCCL
ONCE, after the timed region (`metric_allreduce` in the line), when N>1.
Inputs (graph, feature tables, weights, every batch's id/time arrays) are resident in HBM before the
timed region.  N>1: one process per GPU, graph/tables/weights replicated, whole batches dealt
round-robin (dyglib_amd.distributed.shard_batch_indices: rank r takes batches r, r+N, ...: weak
scaling, K steps per rank), no data-path collective; the synthetic graph is generated once per node (local rank 0 ->
/dev/shm -> the other ranks).  At N>1 the line also carries `secondary.lastfm`: BASELINE config 4's workload (L=512,
P=8) sharded over the N ranks the same way.  Rank 0 prints ONE JSON line.
DYGNN_BENCH_FORCE_DIST=1 sets the process group up at world size 1 too (RCCL rehearsal on a one-GPU box).

Besides the contract fields the line carries (rank 0, N = 1):
  parity     the GPU outputs of the timed steps the CPU leg replays, against the oracle: full-size config-1 parity
             observed by whoever runs this file; the process exits non-zero above the 1e-4 bar of BASELINE.json;
  stages     neighbour-lookup rate (bytes by SURVEY §8d's formula), device metrics, `per_call`: the drop-in rate of
             evaluate_models_utils.py:49-152 — one 200-edge step per launch, numpy inputs, through
             dyglib_amd.evaluate_model_link_prediction(fuse_batches=1) — and `full_span`: all 237 batches of the
             evaluation span through the same loop, mean AP / AUC, per-batch AP / AUC of six batches against the oracle;
  secondary  short runs of BASELINE configs 3 (TGAT), 4 (LastFM-shaped DyGFormer, one GPU's share), 5 (TGN) and of the
             training step, each with value, ms/step, roofline and a warmed CPU sample.
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

# the fp32 MFMA peak and the HBM peak of MI355X_MICROARCH.md.
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_HBM_BPS = 8.0e12
PARITY_TOL = 1e-4           # BASELINE.json north_star: "within 1e-4 for fp32 embeddings"
T_START = time.perf_counter()

WORKLOADS = {
    # name: users, items, edges, L, P, batch, edge features
    "wikipedia": dict(users=8227, items=1000, edges=157474, L=64, P=2, batch=200, edge_feat_kind="normal"),
    "lastfm": dict(users=980, items=1000, edges=1293103, L=512, P=8, batch=200, edge_feat_kind="zeros"),  # BASELINE config 4 shape (128 tokens per pair)
    "tiny": dict(users=300, items=50, edges=12000, L=64, P=2, batch=200, edge_feat_kind="normal"),       # CI / smoke sizes
}
WORKLOAD_LABEL = {"wikipedia": "Wikipedia", "lastfm": "LastFM-shaped (config 4)", "tiny": "tiny"}


def flops_per_pair(S_s: int, S_d: int, P: int, Fn=172, Ft=100, C=50, layers=2) -> float:
    """SURVEY.md §8(d) formula: algorithmic work of one (src,dst,t) pair, multiply-add = 2 flops."""
    S, Dm = S_s + S_d, 4 * C
    T = S // P
    return (2 * Ft * S + S * 2 * (2 * C + 2 * C * C) + T * 2 * C * P * (2 * Fn + Ft + C)
            + layers * (T * (2 * Dm * 3 * Dm + 2 * Dm * Dm + 4 * Dm * 4 * Dm) + 4 * T * T * Dm) + 2 * 2 * Dm * Fn)


def cpu_threads() -> int:
    # torch's default (every logical CPU of the host, 256 on the GPU box) oversubscribes these small ops and is ~6x
    # slower than 16 threads, the box's CPU share for one GPU
    return min(16, os.cpu_count() or 1)


def _timed_cpu(step_fn, first: int, max_steps: int, budget_s: float, warm: int = 1):
    """warm untimed steps, then up to max_steps timed ones within ~budget_s seconds.  Returns (n, seconds, outputs)."""
    for i in range(warm):
        step_fn(first + i)
    outs, n, t0 = [], 0, time.perf_counter()
    STEP_TIMES.clear()
    while n < max_steps:
        t1 = time.perf_counter()
        outs.append(step_fn(first + warm + n))
        STEP_TIMES.append(time.perf_counter() - t1)
        n += 1
        if time.perf_counter() - t0 >= budget_s:
            break
    return n, time.perf_counter() - t0, outs


STEP_TIMES: list = []          # per-step seconds of the last _timed_cpu sample (for the median)


# ======================================================================================================================
# DyGFormer workloads (headline: wikipedia; secondary: lastfm)
# ======================================================================================================================
def build_graph_once_per_node(name: str, share=None):
    """The synthetic graph + feature tables of a workload.  With N > 1 ranks on one node (`share` = (dist, local_rank, tag)) local rank 0
    generates them once and the other ranks map its arrays from /dev/shm (copy-on-write, nothing is written back) instead of
    repeating the host work N times on one box; rank 0 removes the files once every rank has them mapped."""
    wl = WORKLOADS[name]
    gen = lambda: syn.make_bipartite_graph(wl["users"], wl["items"], wl["edges"], seed=0, edge_feat_kind=wl["edge_feat_kind"])
    if share is None:
        return gen()
    import shutil
    dist, local_rank, tag = share
    d = os.path.join("/dev/shm", f"dygnn_bench_{tag}_{name}")
    fields = ("src_node_ids", "dst_node_ids", "node_interact_times", "edge_ids", "labels")
    out = None
    if local_rank == 0:
        out = gen()
        try:
            shutil.rmtree(d, ignore_errors=True)
            os.makedirs(d)
            data, nf, ef = out
            for f in fields:
                np.save(os.path.join(d, f + ".npy"), getattr(data, f))
            np.save(os.path.join(d, "node_feat.npy"), nf)
            np.save(os.path.join(d, "edge_feat.npy"), ef)
        except Exception:               # /dev/shm full or not writable: the other ranks find no files and generate the graph themselves
            shutil.rmtree(d, ignore_errors=True)
    dist.barrier()
    if local_rank != 0:
        try:
            ld = lambda f: np.load(os.path.join(d, f + ".npy"), mmap_mode="c")
            out = (syn.InteractionData(*(ld(f) for f in fields)), ld("node_feat"), ld("edge_feat"))
        except Exception:
            out = gen()
    dist.barrier()
    if local_rank == 0:
        shutil.rmtree(d, ignore_errors=True)
    return out


class DygformerWorkload:
    def __init__(self, name: str, dev, impl: int = 0, share=None):
        from dyglib_amd import DyGFormer, MergeLayer, get_neighbor_sampler
        wl = WORKLOADS[name]
        self.name, self.wl, self.dev = name, wl, dev
        self.B, self.L, self.P = wl["batch"], wl["L"], wl["P"]
        self.data, self.node_feat, self.edge_feat = build_graph_once_per_node(name, share)
        self.params = syn.make_dygformer_params(0, patch_size=self.P)
        self.mparams = syn.make_merge_layer_params(1000)
        self.sampler = get_neighbor_sampler(self.data, "recent", seed=1, device=dev)              # full graph, as in evaluation
        self.model = DyGFormer(self.node_feat, self.edge_feat, self.sampler, time_feat_dim=100, channel_embedding_dim=50, patch_size=self.P,
                               num_layers=2, num_heads=2, dropout=0.1, max_input_sequence_length=self.L, device=dev)
        self.model.load_state_dict({k: torch.from_numpy(v) for k, v in self.params.items()})
        self.merge = MergeLayer(172, 172, 172, 1)
        self.merge.load_state_dict({k: torch.from_numpy(v) for k, v in self.mparams.items()})
        self.model, self.merge = self.model.to(dev).eval(), self.merge.to(dev).eval()
        self.model.impl = impl
        # evaluation span = last 30 % of the interactions (val + test), full batches only
        E, B = self.data.num_interactions, self.B
        self.first = int(E * 0.70)
        self.n_batches = (E - self.first) // B
        neg_rs = np.random.RandomState(2)
        uniq_dst = np.unique(self.data.dst_node_ids)
        self.batches = []
        for i in range(self.n_batches):
            sl = slice(self.first + i * B, self.first + (i + 1) * B)
            self.batches.append((self.data.src_node_ids[sl], self.data.dst_node_ids[sl], syn.random_negative_dst(neg_rs, uniq_dst, B),
                                 self.data.node_interact_times[sl]))
        # device-resident inputs [n_batches, B]
        self.src_all = torch.from_numpy(np.stack([b[0] for b in self.batches])).to(dev)
        self.dst_all = torch.from_numpy(np.stack([b[1] for b in self.batches])).to(dev)
        self.neg_all = torch.from_numpy(np.stack([b[2] for b in self.batches])).to(dev)
        self.t_all = torch.from_numpy(np.stack([b[3] for b in self.batches])).to(dev)

    def describe(self) -> str:
        wl = self.wl
        return (f"DyGFormer link-prediction forward, synthetic {self.name}-shaped graph ({wl['users']}+{wl['items']} nodes, {wl['edges']} edges), "
                f"L={self.L}, P={self.P}, batch={self.B}, 2 layers, 2 heads, C=50; pos+neg calls + MergeLayer+sigmoid per step")


def run_dygformer(wk: DygformerWorkload, steps: int, warmup: int, F: int, rank: int, world: int, dist, n_streams: int = 1, keep: int = 0, prime: int = 12) -> dict:
    """K timed steps of this rank in launches of F steps (the positive and negative calls of F consecutive steps = 2F
    independently padded groups of B pairs form ONE grid).  Returns throughput, the per-launch duration of the hot-path
    call from HIP events recorded on its stream, and the device outputs of the first `keep` timed steps."""
    from dyglib_amd import link_prediction_metrics_device
    dev, B = wk.dev, wk.B
    # batches of this rank, round-robin over the ranks (dyglib_amd/distributed.py), cycled when the run is longer than the span
    mine = list(D.shard_batch_indices(wk.n_batches, rank, world))
    order_h = [mine[k % len(mine)] for k in range(warmup + steps)]
    order = torch.tensor(order_h, device=dev)
    streams = [torch.cuda.Stream(dev) for _ in range(n_streams)] if n_streams > 1 else [torch.cuda.current_stream(dev)]
    metric_accs = [torch.zeros(3, dtype=torch.float64, device=dev) for _ in streams]     # per stream: [sum AUC, sum mean-prob gap, steps]
    labels_full = torch.cat([torch.ones(F, B), torch.zeros(F, B)], dim=1).to(dev)         # evaluate_models_utils.py:143
    kept = []

    def launch(idx: torch.Tensor, li: int, ev=None, keep_out: bool = False):
        """the steps `idx` (batch numbers) as ONE hot-path launch of 2*len(idx) groups"""
        nsteps = idx.numel()
        st = streams[li % len(streams)]
        with torch.cuda.stream(st), torch.no_grad():
            src = wk.src_all[idx]
            srcs = torch.cat([src, src])                       # negative sources = batch sources (evaluate_models_utils.py:62-63)
            dsts = torch.cat([wk.dst_all[idx], wk.neg_all[idx]])
            ts = torch.cat([wk.t_all[idx], wk.t_all[idx]])
            if ev is not None:
                ev[0].record(st)
                wk.model._kernel_events = (ev[2], ev[3])      # recorded by the library right around the fused kernel's launch
            s, d = wk.model.compute_src_dst_node_temporal_embeddings_many(srcs, dsts, ts, pos_neg_halves=True)      # [2n, B, 172]
            if ev is not None:
                wk.model._kernel_events = None
                ev[1].record(st)
            prob = wk.merge.link_probabilities(s.reshape(-1, s.shape[-1]), d.reshape(-1, d.shape[-1])).reshape(2, nsteps, B)
            pos, negp = prob[0], prob[1]
            # per-step ROC AUC on the device (dygnn_link_metrics: evaluate_models_utils.py:139-150 without the host round trip), reduced over RCCL
            predicts = torch.cat([pos, negp], dim=1)
            _, auc, _, _ = link_prediction_metrics_device(predicts, labels_full[:nsteps])
            m = torch.stack([auc.sum(), (pos.mean(dim=1) - negp.mean(dim=1)).double().sum(),
                             torch.full((), float(nsteps), dtype=torch.float64, device=dev)])
            metric_accs[li % len(streams)].add_(m)               # per-rank sums; ONE all-reduce after the timed region (SURVEY §8e needs only the totals)
            if keep_out:
                kept.append((s, d, prob))

    def run_steps(first: int, count: int, evs=None, keep_steps: int = 0):
        li, done = 0, 0
        while done < count:
            n = min(F, count - done)
            launch(order[first + done:first + done + n], li, None if evs is None else evs[li], keep_out=done < keep_steps)
            done += n
            li += 1

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # untimed: `prime` launches of the timed launch shape (first-use workspaces, packed weights; the chip reaches its sustained clock only
    # after tens of milliseconds of load — measured: the same launch runs 5-6 % slower in the first 15 ms of a process's GPU activity),
    # then the W warm-up steps
    for _ in range(max(1, prime)):
        launch(order[:F] if order.numel() >= F else order, 0)
    run_steps(0, warmup)
    sync_all()
    [a.zero_() for a in metric_accs]
    n_launch = (steps + F - 1) // F
    # per timed launch: (call start, call end, kernel start, kernel end).  The kernel pair is recorded by the library on the launch stream
    # immediately before / after the fused forward (dygnn_dygformer_taps.ev_kernel_*); they are created here by a first record().
    events = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(4)) for _ in range(n_launch)]
    for evs in events:
        for e in evs:
            e.record(streams[0])
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run_steps(warmup, steps, events, keep_steps=keep)
    sync_all()
    elapsed = time.perf_counter() - t0
    acc_dev = metric_accs[0] if len(metric_accs) == 1 else torch.stack(metric_accs).sum(dim=0)
    reduce_ms = None
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # the evaluation's only exchange: [sum AUC, sum gap, #steps] of every rank, float64, summed over RCCL — once, outside the timed loop
        t_r = time.perf_counter()
        dist.all_reduce(acc_dev, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize(dev)
        reduce_ms = (time.perf_counter() - t_r) * 1e3
    # dominant kernel = the fused forward, one launch per F steps: mean duration of the FULL launches, from HIP events recorded on the
    # launch stream inside the timed region right around that kernel's launch (the generic path: around the whole call)
    sizes = [min(F, steps - i * F) for i in range(n_launch)]
    full = [i for i in range(n_launch) if sizes[i] == sizes[0]]
    impl_fused = wk.model.impl in (0, 3)
    call_ms = float(np.mean([events[i][0].elapsed_time(events[i][1]) for i in full]))           # window search + fused forward + host gaps
    launch_ms = float(np.mean([events[i][2].elapsed_time(events[i][3]) for i in full])) if impl_fused else call_ms
    acc = acc_dev.cpu().numpy()
    return dict(elapsed=elapsed, value=steps * B * world / elapsed, launch_ms=launch_ms, call_ms=call_ms, steps_per_launch=sizes[0], n_launch=n_launch,
                timed_launches=len(full), mean_auc=float(acc[0] / max(acc[2], 1)), metric_steps=int(round(acc[2])), metric_allreduce_ms=reduce_ms,
                kept=kept, order=order_h[warmup:warmup + steps], streams=len(streams))


def dygformer_roofline(wk: DygformerWorkload, res: dict, impl: int) -> dict:
    B, L, P = wk.B, wk.L, wk.P
    S = ((L + P - 1) // P) * P                      # every batch of these workloads pads to the full window
    fpp = flops_per_pair(S, S, P)
    pairs = 2 * res["steps_per_launch"] * B
    flop_per_launch = fpp * pairs
    achieved = flop_per_launch / (res["launch_ms"] * 1e-3) / 1e12
    # HBM traffic of the fused kernel: bytes per pair from the committed rocprofv3 --pmc passes of this command
    # (tools/pmc_profile.sh -> profiles/*_traffic.json; counters cannot be read from inside the process) x pairs per launch
    traffic, src = None, None
    try:
        import glob
        for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):       # the latest round's file of THIS workload
            tj = json.load(open(tf))
            if wk.name == tj.get("workload", "wikipedia") and impl in (0, 3):
                traffic, src = tj["hbm_bytes_per_pair"] * pairs, os.path.relpath(tf, ROOT)
                break
    except Exception:
        pass
    kern = {1: "generic multi-kernel path"}.get(impl, "k_dygformer_fused3<%d>" % (4 if 2 * ((L + P - 1) // P) <= 64 else 8))
    return {"bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_source": src,
            "kernel": kern, "flop_per_pair": fpp,
            "flop_per_launch": flop_per_launch, "ms_per_launch": round(res["launch_ms"], 4), "ms_per_call": round(res["call_ms"], 4),
            "call": "window-search launches + the kernel + host gaps between them", "pairs_per_launch": pairs,
            "timed_launches": res["timed_launches"]}


# ---- CPU baseline + parity of the headline workload ---------------------------------------------------------------------------
def cpu_baseline_and_parity(wk: DygformerWorkload, res: dict, budget_s: float, max_steps: int, one_thread_s: float = 0.0):
    """The CPU oracle (restatement of the reference path, kind 'port') timed on this host on a bounded sample of the SAME
    workload — the first timed steps of the GPU run — and, from the same replay, the GPU/oracle parity of those steps."""
    from oracle import dygformer_oracle as orc
    cores = cpu_threads()
    torch.set_num_threads(cores)
    data = wk.data
    adj = orc.OracleAdjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    P, L = wk.P, wk.L

    order = list(res["order"])
    order += [b for b in range(wk.n_batches) if b not in set(order)][:max(0, max_steps - len(order))]     # CPU-only steps beyond the GPU's timed ones

    def step(k):                                     # k-th timed step of the GPU run (k = -1: warm-up on another batch)
        src, dst, neg, t = wk.batches[order[k] if k >= 0 else order[-1]]
        with torch.no_grad():
            s, d = orc.dygformer_forward(wk.params, wk.node_feat, wk.edge_feat, adj, src, dst, t, P, L)
            ns, nd = orc.dygformer_forward(wk.params, wk.node_feat, wk.edge_feat, adj, src, neg, t, P, L)
            pos = orc.merge_layer(wk.mparams, s, d).squeeze(-1).sigmoid()
            ng = orc.merge_layer(wk.mparams, ns, nd).squeeze(-1).sigmoid()
        return s, d, ns, nd, pos, ng
    n, el, outs = _timed_cpu(step, -1, min(max_steps, len(order)), budget_s, warm=1)
    med = float(np.median(STEP_TIMES))
    base = {"value": round(wk.B / med, 1), "unit": "edges/s", "cores": cores, "kind": "port", "statistic": "median step time",
            "mean_value": round(n * wk.B / el, 1), "steps": n,
            "sample": f"{n} 200-edge steps of the evaluation span ({el:.1f} s; the GPU's timed steps first) after 1 warm-up step; "
                      f"oracle/dygformer_oracle.py (numpy sampling + PyTorch-CPU fp32 dense ops, torch threads={cores})"}
    if one_thread_s > 0:                             # the same steps at one thread
        torch.set_num_threads(1)
        n1, el1, _ = _timed_cpu(step, -1, 8, one_thread_s, warm=1)
        base["one_thread"] = {"value": round(wk.B / float(np.median(STEP_TIMES)), 1), "unit": "edges/s", "cores": 1, "steps": n1,
                              "sample": f"{n1} of the same steps ({el1:.1f} s), median, torch threads=1"}
        torch.set_num_threads(cores)
    # parity: GPU outputs kept from the timed launches, step k = element k % F of launch k // F
    F = res["steps_per_launch"]
    e_emb = e_prob = ref_emb = 0.0
    checked = 0
    for k in range(min(n, len(res["order"]))):
        if k // F >= len(res["kept"]):
            break
        s, d, prob = res["kept"][k // F]
        j, nst = k % F, s.shape[0] // 2
        os_, od, ons, ond, opos, oneg = outs[k]
        for g, o in ((s[j], os_), (d[j], od), (s[nst + j], ons), (d[nst + j], ond)):
            e_emb = max(e_emb, float((g.cpu() - o).abs().max()))
            ref_emb = max(ref_emb, float(o.abs().max()))
        e_prob = max(e_prob, float((prob[0, j].cpu() - opos).abs().max()), float((prob[1, j].cpu() - oneg).abs().max()))
        checked += 1
    tol_emb = PARITY_TOL * max(1.0, ref_emb)
    parity = {"steps": checked, "pairs": checked * 2 * wk.B, "max_abs_prob": e_prob, "max_abs_emb": e_emb, "max_abs_ref_emb": round(ref_emb, 3),
              "tolerance": PARITY_TOL, "ok": bool(checked > 0 and e_prob <= PARITY_TOL and e_emb <= tol_emb),
              "against": "oracle/dygformer_oracle.py on the same timed steps (full-size graph); embeddings within 1e-4*max(1,max|ref|), probabilities within 1e-4"}
    return base, parity


# ======================================================================================================================
# stages
# ======================================================================================================================
def sampler_stage(sampler, data, dev, n_queries: int = 2_000_000, k: int = 20, reps: int = 3) -> dict:
    """get_historical_neighbors ('recent', k = 20) on random (endpoint, time) queries of the evaluation span: queries/s and the
    algorithmic-byte rate against the 8 TB/s HBM peak (the reference: 149 k queries/s on the CPU).  2 M queries = 1.5 GB of algorithmic
    bytes per pass, six times the 256 MB Infinity Cache: the outputs stream to HBM.  Bytes per query by SURVEY §8d's
    formula on this library's CSR (16 B per entry: int32 id, int32 edge id, float64 time): 8*ceil(log2(deg+1)) binary-search probe
    bytes + 16*min(history, k) window bytes + 20*k output bytes (int64, int64, float32) + 16 query bytes."""
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
        out = sampler.get_historical_neighbors_device(nodes, times, k)
    e1.record()
    torch.cuda.synchronize(dev)
    sec = e0.elapsed_time(e1) * 1e-3 / reps
    deg = (sampler.csr.indptr[nodes_h + 1] - sampler.csr.indptr[nodes_h]).astype(np.float64)
    hist = (out[0] != 0).sum(dim=1).cpu().numpy().astype(np.float64)            # neighbours found per query (ids start at 1)
    algo = float((8 * np.ceil(np.log2(deg + 1))).sum() + 16 * hist.sum() + 20.0 * k * len(nodes_h) + 16 * len(nodes_h))
    return {"sampler_recent_k20_queries_per_s": round(len(nodes_h) / sec), "sampler_algorithmic_bytes_per_query": round(algo / len(nodes_h), 1),
            "sampler_algorithmic_GBps": round(algo / sec / 1e9, 1), "sampler_frac_of_hbm_peak": round(algo / sec / PEAK_HBM_BPS, 4),
            "queries": len(nodes_h)}


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


def per_call_stage(wk: DygformerWorkload, n_batches: int = 60) -> dict:
    """The drop-in caller's rate (reference evaluate_models_utils.py:49-152 unchanged): `evaluate_model_link_prediction` over
    n_batches 200-edge batches of the evaluation span with fuse_batches=1 — one step per launch, numpy id/time arrays handed
    over per batch (PCIe-inclusive), negatives drawn by the seeded `random` NegativeEdgeSampler, device AP/AUC/BCE, one
    synchronisation at the end."""
    import torch.nn as nn
    from dyglib_amd import NegativeEdgeSampler, evaluate_model_link_prediction, get_idx_data_loader
    data, B = wk.data, wk.B
    n_batches = min(n_batches, wk.n_batches)
    idx = list(range(wk.first, wk.first + n_batches * B))
    loader = get_idx_data_loader(idx, batch_size=B, shuffle=False)
    negs = NegativeEdgeSampler(data.src_node_ids, data.dst_node_ids, seed=0)
    model = nn.Sequential(wk.model, wk.merge)
    run = lambda: evaluate_model_link_prediction("DyGFormer", model, wk.sampler, loader, negs, data, nn.BCELoss(), fuse_batches=1)
    run()                                                   # warm-up pass (same shapes)
    torch.cuda.synchronize(wk.dev)
    t0 = time.perf_counter()
    losses, metrics = run()
    torch.cuda.synchronize(wk.dev)
    el = time.perf_counter() - t0
    return {"per_call": {"value": round(n_batches * B / el, 1), "unit": "edges/s", "ms_per_step": round(el / n_batches * 1e3, 4), "steps": n_batches,
                         "what": "dyglib_amd.evaluate_model_link_prediction(fuse_batches=1): one 200-edge step per launch, numpy inputs per batch, "
                                 "device metrics, one host synchronisation per evaluation",
                         "mean_auc": round(float(np.mean([m["roc_auc"] for m in metrics])), 4)}}


def full_span_stage(wk: DygformerWorkload, oracle_batches: int = 6, oracle_all: bool = False) -> dict:
    """The metric over the span it is defined on: ALL batches of the last 30 % of the interactions (236 full batches + the
    43-edge tail = 237) through `evaluate_model_link_prediction` — the loop of evaluate_models_utils.py:49-152 with its seeded `random`
    negative sampler, per-batch AP / AUC / BCE on the device, host arrays in, one synchronisation at the end — and the per-batch AP / AUC of
    `oracle_batches` of them (first, last = the ragged tail, and evenly spaced ones; all 237 with --full-span-oracle) recomputed by the CPU
    oracle on the same negative draws."""
    import torch.nn as nn
    from dyglib_amd import NegativeEdgeSampler, evaluate_model_link_prediction, get_idx_data_loader
    data, B = wk.data, wk.B
    idx = list(range(wk.first, data.num_interactions))
    nb = (len(idx) + B - 1) // B
    model = nn.Sequential(wk.model, wk.merge)
    mk = lambda: (get_idx_data_loader(idx, batch_size=B, shuffle=False), NegativeEdgeSampler(data.src_node_ids, data.dst_node_ids, seed=0))
    run = lambda: evaluate_model_link_prediction("DyGFormer", model, wk.sampler, *mk(), data, nn.BCELoss())
    run()                                                    # warm-up pass (same shapes)
    torch.cuda.synchronize(wk.dev)
    t0 = time.perf_counter()
    losses, metrics = run()
    torch.cuda.synchronize(wk.dev)
    el = time.perf_counter() - t0
    out = {"value": round(len(idx) / el, 1), "unit": "edges/s", "batches": nb, "edges": len(idx), "seconds": round(el, 4),
           "mean_average_precision": float(np.mean([m["average_precision"] for m in metrics])), "mean_roc_auc": float(np.mean([m["roc_auc"] for m in metrics])),
           "mean_loss": float(np.mean(losses)),
           "what": "dyglib_amd.evaluate_model_link_prediction over the whole evaluation span (evaluate_models_utils.py:49-152): numpy inputs per batch, "
                   "32 batches per launch, device AP / AUC / BCE, one host synchronisation"}
    if oracle_batches > 0 or oracle_all:
        from oracle import dygformer_oracle as orc, metrics_oracle as mo
        torch.set_num_threads(cpu_threads())
        adj = orc.OracleAdjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
        negs = NegativeEdgeSampler(data.src_node_ids, data.dst_node_ids, seed=0)
        negs.reset_random_state()
        pick = set(range(nb)) if oracle_all else set(np.unique(np.linspace(0, nb - 1, oracle_batches).round().astype(int)).tolist())
        d_ap = d_auc = 0.0
        o_ap, o_auc = [], []
        t1 = time.perf_counter()
        for b in range(nb):
            sl = idx[b * B:(b + 1) * B]
            _, neg = negs.sample(size=len(sl))               # every batch draws, picked or not: the stream stays the evaluation's
            if b not in pick:
                continue
            src, dst, t = data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl]
            with torch.no_grad():
                pos, ng = orc.link_prediction_step(wk.params, wk.mparams, wk.node_feat, wk.edge_feat, adj, src, dst, neg, t, wk.P, wk.L)
            y = np.concatenate([np.ones(len(sl)), np.zeros(len(sl))])
            pr = np.concatenate([pos.numpy(), ng.numpy()])
            o_ap.append(mo.average_precision(y, pr)), o_auc.append(mo.roc_auc(y, pr))
            d_ap = max(d_ap, abs(o_ap[-1] - metrics[b]["average_precision"]))
            d_auc = max(d_auc, abs(o_auc[-1] - metrics[b]["roc_auc"]))
        out["oracle_check"] = {"batches": sorted(pick) if not oracle_all else f"all {nb}", "max_abs_ap_diff": d_ap, "max_abs_auc_diff": d_auc,
                               "oracle_mean_ap_of_checked": float(np.mean(o_ap)), "oracle_mean_auc_of_checked": float(np.mean(o_auc)),
                               "gpu_mean_ap_of_checked": float(np.mean([metrics[b]["average_precision"] for b in sorted(pick)])),
                               "gpu_mean_auc_of_checked": float(np.mean([metrics[b]["roc_auc"] for b in sorted(pick)])),
                               "seconds": round(time.perf_counter() - t1, 1),
                               "ok": bool(d_ap <= 1e-3 and d_auc <= 1e-3),
                               "against": "oracle/dygformer_oracle.py + oracle/metrics_oracle.py on the same batches and negative draws; AP / AUC are rank "
                                          "statistics of 400 scores each within 1e-4 of the oracle's: a swap of two near-tied scores moves them by ~1/40000, bar 1e-3"}
    return {"full_span": out}


# ======================================================================================================================
# secondary workloads (BASELINE configs 3, 4, 5 and the training step); each returns one dict
# ======================================================================================================================
def bench_lastfm(dev, steps: int = 16, warmup: int = 8, F: int = 8, cpu_budget_s: float = 10.0, cpu_max_steps: int = 30) -> dict:
    """BASELINE config 4's shape on one GPU (its 8-GPU form is `--workload lastfm --gpus 8`): L=512, P=8 -> 128 tokens per pair."""
    wk = DygformerWorkload("lastfm", dev)
    _prime_gpu(dev)               # the leg follows the headline's CPU sample (seconds of host-only work): clock ramp
    res = run_dygformer(wk, steps, warmup, F, 0, 1, None, keep=min(steps, cpu_max_steps) if cpu_budget_s > 0 else 0)
    out = {"metric": "edges/sec (link-prediction fwd) DyGFormer LastFM-shaped (config 4), 1 GPU", "value": round(res["value"], 1), "unit": "edges/s",
           "steps": steps, "warmup": warmup, "ms_per_step": round(res["elapsed"] / steps * 1e3, 4), "steps_per_launch": res["steps_per_launch"],
           "config": {"workload": wk.describe()}, "roofline": dygformer_roofline(wk, res, 0)}
    if cpu_budget_s > 0:
        out["cpu_baseline"], out["parity"] = cpu_baseline_and_parity(wk, res, cpu_budget_s, cpu_max_steps)
    return out


def bench_tgat(dev, steps: int = 192, warmup: int = 3, fuse_steps: int = 32, edges: int = 672447, cpu_budget_s: float = 20.0,
               cpu_max_steps: int = 30, one_step_calls: int = 20, uniform_steps: int = 12, larger_calls=(128,)) -> dict:
    """BASELINE config 3: TGAT link-prediction forward, Reddit-shaped synthetic graph (10,000 + 984 nodes, 672,447 edges), k = 20,
    2 layers, batch 200: pos call + neg call + MergeLayer+sigmoid per step.  Rows do not depend on the batch they are in (fixed k,
    no batch-dependent padding), so F steps are one call on F*200 edges; the one-step-per-call rate is reported beside it."""
    from dyglib_amd import TGAT, MergeLayer, get_neighbor_sampler
    B, K = 200, 20
    data, nf, ef = syn.make_bipartite_graph(10000, 984, edges, seed=0)
    params, mparams = syn.make_tgat_params(0), syn.make_merge_layer_params(1000)
    sampler = get_neighbor_sampler(data, "recent", seed=1, device=dev)
    model = TGAT(nf, ef, sampler, 100, num_layers=2, num_heads=2, dropout=0.1, device=dev)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    merge = MergeLayer(172, 172, 172, 1)
    merge.load_state_dict({k: torch.from_numpy(v) for k, v in mparams.items()})
    model, merge = model.to(dev).eval(), merge.to(dev).eval()
    E = data.num_interactions
    first = int(E * 0.7)
    nb = (E - first) // B
    rs, ud = np.random.RandomState(2), np.unique(data.dst_node_ids)
    F = max(1, min(fuse_steps, steps))
    steps = (steps + F - 1) // F * F
    n_calls = min(nb // F, steps // F + warmup)
    host = []
    for i in range(n_calls):
        sl = slice(first + i * B * F, first + (i + 1) * B * F)
        host.append((data.src_node_ids[sl], data.dst_node_ids[sl], syn.random_negative_dst(rs, ud, B * F), data.node_interact_times[sl]))
    batches = [tuple(torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in h) for h in host]

    def call(s, d, n, t):
        with torch.no_grad():
            # the positive and the negative call as ONE call on the roots [sources ; destinations ; negative destinations] (rows do not depend
            # on the batch they are in; the negative call's sources are the positive call's): the shared source side is computed once, and so is
            # every other repeated (node, time) entry of level 1 (level de-duplication, tgat.hip)
            se, de, ne = model.compute_step_embeddings(s, d, n, t, num_neighbors=K)
            p = merge.link_probabilities(torch.cat([se, se]), torch.cat([de, ne]))
            return p[:len(s)], p[len(s):]
    _prime_gpu(dev)               # the leg follows seconds of host-only work (graph construction, the previous leg's CPU sample): clock ramp
    for i in range(warmup):
        call(*batches[i % len(batches)])
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps // F):
        keep = call(*batches[(warmup + i) % len(batches)])
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    total_entries, computed_entries = model.last_level_entries()          # of the last call (F steps, positive and negative together)
    # one step per call (the reference caller's shape): the first `one_step_calls` steps of the last fused call
    s, d, n, t = batches[(warmup + steps // F - 1) % len(batches)]
    one = [(s[j * B:(j + 1) * B], d[j * B:(j + 1) * B], n[j * B:(j + 1) * B], t[j * B:(j + 1) * B]) for j in range(max(1, min(F, one_step_calls)))]
    call(*one[0])
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    outs1 = [call(*o) for o in one]
    torch.cuda.synchronize(dev)
    el1 = (time.perf_counter() - t1) / len(one)
    # executed work: 1.062 MFLOP per COMPUTED (node, time) entry (q 2*272^2 + W_k^T q 2*2*136*444 + W_v z 2*2*444*136 + residual_fc 2*272^2 +
    # merge fc1 2*444*172 + fc2 2*172^2 + scores and weighted sums 2*2*20*444*2; K/V never materialised).  The reference
    # computes 2*2*200*(1+21) = 17,600 entries per step at 10.19 MFLOP each (179.4 GFLOP per step).
    per_step, sec_step = computed_entries / F, el / steps
    out = {"metric": "edges/sec (link-prediction fwd) TGAT Reddit-shaped (config 3)", "value": round(steps * B / el, 1), "unit": "edges/s",
           "steps": steps, "warmup": warmup * F, "ms_per_step": round(sec_step * 1e3, 4), "steps_per_call": F,
           "one_step_per_call": {"value": round(B / el1, 1), "unit": "edges/s", "ms_per_step": round(el1 * 1e3, 4), "steps": len(one)},
           "config": {"workload": f"TGAT link-prediction forward, synthetic Reddit-shaped graph (10000+984 nodes, {edges} edges), k=20, 2 layers, batch=200"},
           "roofline": {"bound": "mfma", "achieved": round(1.061952e6 * per_step / sec_step / 1e12, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(1.061952e6 * per_step / sec_step / (PEAK_F32_MFMA_TFLOPS * 1e12), 4), "traffic": None,
                        "entries_per_step": {"reference": 2 * 2 * B * (1 + K + 1), "in_call_layout": total_entries / F, "computed": round(per_step, 1)},
                        "reference_formulation_equivalent_TFLOPs": round(179.4e9 / sec_step / 1e12, 1),
                        "note": "executed flops of the computed (node, time) entries (1.062 MFLOP each; duplicates of level 1 computed once)"}}
    # the reference's best TGAT configuration on Reddit samples `uniform` (utils/load_configs.py:83-84): the draws are the sampler's numpy
    # RandomState stream, replayed on the host (one library call per level: dygnn_mt19937_choice_rows_host), two calls per step like the
    # reference (evaluate_models_utils.py:126-136).  Parity of this path: the reference's fixtures (tests/golden/tgat_rand_*, sampling_*).
    # Larger calls: rows do not depend on the batch they are in, so the caller may fuse any number of steps; the more steps a call holds, the
    # more of its level-1 (node, time) entries repeat and are computed once (at 128 steps: 2.5 k of the reference's 17.6 k entries per step).
    # Memory is what bounds it (level arrays + layer buffers: ~23 GB at 128 steps — a tenth of the 288 GB).  The headline above stays at 32 steps
    # per call (the figure of the earlier rounds); evaluate_model_link_prediction takes the same choice through `fuse_batches`.
    if larger_calls:
        out["larger_calls"] = {}
        for F2 in larger_calls:
            n2 = min(3, nb // F2 - 1)
            if n2 < 1:
                continue
            hb = []
            for i in range(n2 + 1):
                sl = slice(first + i * B * F2, first + (i + 1) * B * F2)
                hb.append(tuple(torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in
                                (data.src_node_ids[sl], data.dst_node_ids[sl], syn.random_negative_dst(rs, ud, B * F2), data.node_interact_times[sl])))
            call(*hb[0])
            torch.cuda.synchronize(dev)
            t3 = time.perf_counter()
            for i in range(n2):
                call(*hb[1 + i])
            torch.cuda.synchronize(dev)
            el3 = (time.perf_counter() - t3) / (n2 * F2)
            tot3, comp3 = model.last_level_entries()
            out["larger_calls"][str(F2)] = {"value": round(B / el3, 1), "unit": "edges/s", "ms_per_step": round(el3 * 1e3, 4), "calls": n2,
                                            "computed_entries_per_step": round(comp3 / F2, 1)}
            del hb
            torch.cuda.empty_cache()
    if uniform_steps <= 0:           # profiling runs (tools/bench_tgat.py --plain): the fused `recent` calls only
        return out
    usampler = get_neighbor_sampler(data, "uniform", seed=3, device=dev)
    model.set_neighbor_sampler(usampler)
    hs_, hd_, hn_, ht_ = host[0]
    ustep = lambda j: (merge.link_probabilities(*model.compute_src_dst_node_temporal_embeddings(hs_[j * B:(j + 1) * B], hd_[j * B:(j + 1) * B], ht_[j * B:(j + 1) * B], num_neighbors=K)),
                       merge.link_probabilities(*model.compute_src_dst_node_temporal_embeddings(hs_[j * B:(j + 1) * B], hn_[j * B:(j + 1) * B], ht_[j * B:(j + 1) * B], num_neighbors=K)))
    nu = min(uniform_steps, F)
    with torch.no_grad():
        ustep(0)
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        for j in range(nu):
            ustep(j)
        torch.cuda.synchronize(dev)
    elu = (time.perf_counter() - t2) / nu
    out["uniform"] = {"value": round(B / elu, 1), "unit": "edges/s", "ms_per_step": round(elu * 1e3, 3), "steps": nu,
                      "what": "sample_neighbor_strategy='uniform' (the reference's best TGAT configuration on Reddit): RandomState draws replayed on the host, "
                              "positive and negative call per step, PCIe-inclusive; parity by the reference fixtures tests/golden/tgat_rand_*.npz"}
    model.set_neighbor_sampler(sampler)
    if cpu_budget_s > 0:
        from oracle import dygformer_oracle as orc, tgat_oracle as torc
        torch.set_num_threads(cpu_threads())
        adj = orc.OracleAdjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
        nft, eft = torch.from_numpy(nf), torch.from_numpy(ef)
        tp = {kk: torch.from_numpy(v) for kk, v in params.items()}
        hs, hd, hn, ht = (x.cpu().numpy() for x in (s, d, n, t))
        mp = mparams

        def cstep(j):
            sl = slice((j % F) * B, (j % F + 1) * B)
            with torch.no_grad():
                a, b_ = torc.tgat_forward(tp, nft, eft, adj, hs[sl], hd[sl], ht[sl], 2, K, 2)
                c, e = torc.tgat_forward(tp, nft, eft, adj, hs[sl], hn[sl], ht[sl], 2, K, 2)
                return orc.merge_layer(mp, a, b_).squeeze(-1).sigmoid(), orc.merge_layer(mp, c, e).squeeze(-1).sigmoid()
        ncpu, cel, couts = _timed_cpu(cstep, -1, min(cpu_max_steps, F), cpu_budget_s, warm=1)
        out["cpu_baseline"] = {"value": round(ncpu * B / cel, 1), "unit": "edges/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{ncpu} of the same steps ({cel:.1f} s) after 1 warm-up step, oracle/tgat_oracle.py"}
        err = 0.0
        for j in range(min(ncpu, len(outs1))):
            err = max(err, float((outs1[j][0].cpu() - couts[j][0]).abs().max()), float((outs1[j][1].cpu() - couts[j][1]).abs().max()))
        fused_err = 0.0
        for j in range(min(ncpu, F)):
            fused_err = max(fused_err, float((keep[0][j * B:(j + 1) * B].cpu() - couts[j][0]).abs().max()),
                            float((keep[1][j * B:(j + 1) * B].cpu() - couts[j][1]).abs().max()))
        out["parity"] = {"steps": min(ncpu, len(outs1)), "max_abs_prob": max(err, fused_err), "tolerance": PARITY_TOL, "ok": bool(max(err, fused_err) <= PARITY_TOL),
                         "against": "oracle/tgat_oracle.py, link probabilities of the same steps (one-step calls and the fused call)"}
    return out


def _prime_gpu(dev, seconds: float = 0.4) -> None:
    """Unrelated GPU work (fp32 matmuls) for `seconds`: brings the clocks up before a short timed region that follows host-only work."""
    a = torch.randn(2048, 2048, device=dev)
    t = time.perf_counter()
    while time.perf_counter() - t < seconds:
        for _ in range(8):
            a = (a @ a) * 1e-3
        torch.cuda.synchronize(dev)


def bench_tgn(dev, steps: int = 100, warmup: int = 60, cpu_budget_s: float = 10.0, cpu_max_steps: int = 30, two_calls: bool = False) -> dict:
    """BASELINE config 5: TGN link-prediction forward on a MOOC-shaped synthetic graph (7,047 + 97 nodes, 411,749 edges, 4 non-zero
    edge-feature columns), k = 10, 1 layer, batch 200, batches strictly in chronological order from interaction 0: negative call +
    positive call (memory update) + MergeLayer+sigmoid per step.  TGN does not shard: "replicas only" (SURVEY.md §8e)."""
    from dyglib_amd import MemoryModel, MergeLayer, get_neighbor_sampler
    B, K = 200, 10
    data, nf, ef = syn.make_bipartite_graph(7047, 97, 411749, seed=0, edge_feat_kind="sparse4")
    params, mparams = syn.make_tgn_params(0, nf.shape[0], num_layers=1), syn.make_merge_layer_params(1000)
    sampler = get_neighbor_sampler(data, "recent", seed=1, device=dev)
    model = MemoryModel(nf, ef, sampler, 100, model_name="TGN", num_layers=1, num_heads=2, dropout=0.1, device=dev)
    sd = model.state_dict()
    sd.update({k: torch.from_numpy(v) for k, v in params.items()})
    model.load_state_dict(sd)
    merge = MergeLayer(172, 172, 172, 1)
    merge.load_state_dict({k: torch.from_numpy(v) for k, v in mparams.items()})
    model, merge = model.to(dev).eval(), merge.to(dev).eval()
    rs, ud = np.random.RandomState(2), np.unique(data.dst_node_ids)
    n = steps + warmup
    host = [(data.src_node_ids[i * B:(i + 1) * B], data.dst_node_ids[i * B:(i + 1) * B], syn.random_negative_dst(rs, ud, B),
             data.node_interact_times[i * B:(i + 1) * B], data.edge_ids[i * B:(i + 1) * B]) for i in range(n)]
    batches = [tuple(torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in b) for b in host]
    # the step as ONE resident batch [positives ; negatives] (what compute_step_embeddings builds from its four id arrays)
    joint = [tuple(torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (np.concatenate([s_, s_]), np.concatenate([d_, n_]), np.concatenate([t_, t_]), e_))
             for s_, d_, n_, t_, e_ in host]

    def step(i):
        s, d, ng, t, e = batches[i]
        with torch.no_grad():
            if two_calls:       # the reference's call pattern: negative call, then positive call (evaluate_models_utils.py:85-107)
                a, b_ = model.compute_src_dst_node_temporal_embeddings(s, ng, t, edge_ids=None, edges_are_positive=False, num_neighbors=K)
                c, f = model.compute_src_dst_node_temporal_embeddings(s, d, t, edge_ids=e, edges_are_positive=True, num_neighbors=K)
                return merge.link_probabilities(c, f), merge.link_probabilities(a, b_)
            s2, d2, t2, e2 = joint[i]         # both calls as one (same state, written once at the end); no per-step concatenation kernels
            se, de = model.compute_step_embeddings_joint(s2, d2, t2, e2, B, num_neighbors=K)
            p = merge.link_probabilities(se, de)
            return p[:B], p[B:]
    model.memory_bank.__init_memory_bank__()
    _prime_gpu(dev)               # the 100 timed steps are 17 ms of GPU work: without this they would run inside the clock ramp after the host-only setup
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    gouts = [step(warmup + i) for i in range(steps)]
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    sec_step = el / steps
    # executed flops per step (linear attention form), R = 3B root entries (src, dst, negative dst; the negative call's
    # sources are the batch sources): per root q 2*272^2 + W_k^T q 2*2*136*444 + W_v z 2*2*444*136 + residual_fc 2*272^2 + merge
    # (2*444*172 + 2*172^2) + scores / weighted sums 2*2*k*444*2; GRU 2*3*172*(616+172) per updated node (<= the 3B*(1+k) nodes read).
    # Algorithmic bytes per step: R*k neighbour rows (memory 688 + edge 688 B) + R*k CSR entries (16 B) + the updated nodes' message (2464 B)
    # and memory rows (688 B read + write) + 2*R outputs (688 B).
    R = 3 * B
    flop = R * (2 * 272 * 272 * 2 + 4 * 2 * 136 * 444 + 2 * 444 * 172 + 2 * 172 * 172 + 8 * K * 444) + 2 * B * 2 * 3 * 172 * 788
    byts = R * K * (688 + 688 + 16) + 2 * B * (2464 + 3 * 688) + R * 688
    out = {"metric": "edges/sec (link-prediction fwd) TGN MOOC-shaped (config 5)", "value": round(steps * B / el, 1), "unit": "edges/s",
           "steps": steps, "warmup": warmup, "ms_per_step": round(sec_step * 1e3, 4), "scaling": "replicas only",
           "config": {"workload": "TGN link-prediction forward, synthetic MOOC-shaped graph (7047+97 nodes, 411749 edges), k=10, 1 layer, batch=200, sequential batches",
                      "calls_per_step": 2 if two_calls else 1},
           "roofline": {"bound": "hbm", "achieved": round(byts / sec_step / 1e9, 2), "peak": PEAK_HBM_BPS / 1e9, "unit": "GB/s",
                        "frac": round(byts / sec_step / PEAK_HBM_BPS, 5), "traffic": None, "bytes_per_step": byts, "flop_per_step": flop,
                        "mfma_frac": round(flop / sec_step / (PEAK_F32_MFMA_TFLOPS * 1e12), 5),
                        "note": f"one 200-edge step is {byts / 1e6:.1f} MB and {flop / 1e9:.2f} GFLOP: {byts / PEAK_HBM_BPS * 1e6:.1f} us of HBM time, "
                                f"{flop / (PEAK_F32_MFMA_TFLOPS * 1e12) * 1e6:.1f} us of MFMA time.  The step is six dependent launches (expansion, "
                                "list + weight packing, GRU, query / key chain, attention + value .. MergeLayer chain, commit) + the link predictor; the three "
                                "row-block chains re-stream the layer's weights per 4-row block from L2 (what bounds them, DESIGN.md 4.6); both fractions "
                                "are reported, neither is close to a roofline"}}
    if cpu_budget_s > 0:
        from oracle import dygformer_oracle as orc, tgn_oracle as tn
        torch.set_num_threads(cpu_threads())
        adj = orc.OracleAdjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
        nft, eft = torch.from_numpy(nf), torch.from_numpy(ef)
        tp = {kk: torch.from_numpy(v) for kk, v in params.items()}
        st = tn.TgnState(nf.shape[0], 172)

        def cstep(i):
            s, d, ng, t, e = host[i]
            with torch.no_grad():
                a, b_ = tn.tgn_forward(tp, nft, eft, adj, st, s, ng, t, None, False, 1, K, 2)
                c, f = tn.tgn_forward(tp, nft, eft, adj, st, s, d, t, e, True, 1, K, 2)
                return orc.merge_layer(mparams, c, f).squeeze(-1).sigmoid(), orc.merge_layer(mparams, a, b_).squeeze(-1).sigmoid()
        # the state is sequential: the CPU replays the warm-up steps untimed, then times the same steps the GPU timed
        ncpu, cel, couts = _timed_cpu(cstep, 0, min(cpu_max_steps, steps), cpu_budget_s, warm=warmup)
        out["cpu_baseline"] = {"value": round(ncpu * B / cel, 1), "unit": "edges/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"the first {ncpu} timed steps ({cel:.1f} s) after replaying the {warmup} warm-up steps, oracle/tgn_oracle.py"}
        err = 0.0
        for j in range(ncpu):
            err = max(err, float((gouts[j][0].cpu() - couts[j][0]).abs().max()), float((gouts[j][1].cpu() - couts[j][1]).abs().max()))
        out["parity"] = {"steps": ncpu, "max_abs_prob": err, "tolerance": PARITY_TOL, "ok": bool(err <= PARITY_TOL),
                         "against": "oracle/tgn_oracle.py: link probabilities of sequential steps (memory carried across them)"}
    return out


def bench_train(dev, steps: int = 20, warmup: int = 8, separate_calls: bool = False, cpu_budget_s: float = 12.0, cpu_max_steps: int = 30,
                shape: str = "wikipedia") -> dict:
    """Training-step throughput of the DyGFormer path: train_link_prediction.py:229-257 in miniature on the
    Wikipedia-shaped workload — positive + negative call in train mode (dropout 0.1), MergeLayer, BCE, backward, Adam step.
    shape="lastfm": the same step at BASELINE config 4's shape (L=512, P=8: 128 tokens per pair, the <8> kernels), GPU timing only."""
    from dyglib_amd import DyGFormer, MergeLayer, get_neighbor_sampler
    wl = WORKLOADS[shape]
    B, L, P = wl["batch"], wl["L"], wl["P"]
    data, nf, ef = syn.make_bipartite_graph(wl["users"], wl["items"], wl["edges"], seed=0, edge_feat_kind=wl["edge_feat_kind"])
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
    rs, uniq_dst = np.random.RandomState(2), np.unique(data.dst_node_ids)
    cache = {}

    def batch(i):
        if i not in cache:
            sl = slice(first + i * B, first + (i + 1) * B)
            cache[i] = (data.src_node_ids[sl], data.dst_node_ids[sl], syn.random_negative_dst(rs, uniq_dst, B), data.node_interact_times[sl])
        return cache[i]

    def step(i):
        nonlocal separate_calls
        src, dst, neg, t = batch(i)
        if separate_calls:      # the reference's call pattern (train_link_prediction.py:229-239), two passes
            ps, pd = model.compute_src_dst_node_temporal_embeddings(src, dst, t)
            ns, nd = model.compute_src_dst_node_temporal_embeddings(src, neg, t)
        else:                   # both calls as one set: one pass when they pad to the same lengths
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
    torch.cuda.synchronize(dev)
    # clock ramp: after seconds of host-only work (the previous legs' CPU baselines, this leg's graph construction) the first few hundred
    # milliseconds of GPU work run at a fraction of the sustained clocks (measured: 7.5 instead of 2.8 ms per step over the first 150 ms);
    # untimed steps until 0.5 s of GPU activity have passed, like the headline's priming launches
    plain = bool(os.environ.get("DYGNN_BENCH_TRAIN_PLAIN"))      # tools/prof_train.sh: exactly warmup + steps one-pass steps under the profiler
    t_prime = time.perf_counter()
    while not plain and time.perf_counter() - t_prime < 0.5:
        for i in range(warmup):
            step(i)
        torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(steps):
        loss = step(warmup + i)
    e1.record()
    torch.cuda.synchronize(dev)
    sec = (time.perf_counter() - t0) / steps
    gpu_ms = e0.elapsed_time(e1) / steps          # main-stream span: equals the wall time when the GPU, not the host, sets the pace
    # forward + backward = 3x the forward's algorithmic flops (each product has two gradient products): 3 * 2B pairs * 137.2 MFLOP
    flop = 3 * 2 * B * flops_per_pair(L, L, P)
    if shape != "wikipedia":          # the long-window shape: the step's rate and fraction only (its CPU autograd step takes tens of seconds)
        return {"value": round(B / sec, 1), "unit": "edges/s", "ms_per_step": round(sec * 1e3, 4), "gpu_stream_ms_per_step": round(gpu_ms, 4), "steps": steps,
                "config": {"workload": f"DyGFormer training step, synthetic {shape}-shaped graph, L={L}, P={P}, batch={B}, dropout 0.1, Adam; pos+neg as one pass"},
                "roofline": {"bound": "mfma", "achieved": round(flop / sec / 1e12, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(flop / sec / (PEAK_F32_MFMA_TFLOPS * 1e12), 4), "flop_per_step": flop}, "final_loss": round(float(loss.detach()), 4)}
    out = {"metric": "edges/sec (link-prediction TRAIN step: fwd pos+neg, bwd, Adam) DyGFormer Wikipedia-shaped", "value": round(B / sec, 1),
           "unit": "edges/s", "ms_per_step": round(sec * 1e3, 4), "gpu_stream_ms_per_step": round(gpu_ms, 4), "steps": steps, "warmup": warmup, "priming": "0.5 s of untimed steps (clock ramp)", "dropout": 0.1, "final_loss": round(float(loss.detach()), 4),
           "config": {"workload": "DyGFormer training step, synthetic wikipedia-shaped graph, L=64, P=2, batch=200, dropout 0.1, Adam",
                      "calls": "two calls" if separate_calls else "pos+neg as one pass"},
           "roofline": {"bound": "mfma", "achieved": round(flop / sec / 1e12, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(flop / sec / (PEAK_F32_MFMA_TFLOPS * 1e12), 4), "traffic": None, "flop_per_step": flop,
                        "note": "3 x the forward's algorithmic flops over the whole step time (host work, optimizer and link predictor included)"}}
    if not separate_calls and not plain:
        # the reference's own call pattern (train_link_prediction.py:229-239: two calls per step) beside the one-pass form, same model state
        separate_calls = True
        for i in range(3):
            step(warmup + steps + i)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for i in range(10):
            step(warmup + steps + 3 + i)
        torch.cuda.synchronize(dev)
        sec2 = (time.perf_counter() - t1) / 10
        separate_calls = False
        out["two_calls_per_step"] = {"value": round(B / sec2, 1), "unit": "edges/s", "ms_per_step": round(sec2 * 1e3, 4), "steps": 10,
                                     "what": "the same step with the reference's two separate forward calls (every kernel on half the grid)"}
        del model, opt
        torch.cuda.empty_cache()
        try:      # the L=512 / P=8 training shape (k_dygformer_fused3<8, true>, k_attn_bwd<8>, k_ffn_bwd<8>)
            out["lastfm_shape"] = bench_train(dev, steps=6, warmup=3, cpu_budget_s=0.0, shape="lastfm")
        except Exception as e:
            out["lastfm_shape"] = {"error": f"{type(e).__name__}: {e}"}
    if cpu_budget_s > 0:
        from oracle import dygformer_oracle as orc
        torch.set_num_threads(cpu_threads())
        cp = {kk: torch.from_numpy(v.copy()).requires_grad_(True) for kk, v in params.items()}
        cm = {kk: torch.from_numpy(v.copy()).requires_grad_(True) for kk, v in mparams.items()}
        adj = orc.OracleAdjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
        copt = torch.optim.Adam(list(cp.values()) + list(cm.values()), lr=1e-4)

        def cstep(i):
            src, dst, neg, t = batch(i)
            ps, pd = orc.dygformer_forward(cp, nf, ef, adj, src, dst, t, P, L)
            ns, nd = orc.dygformer_forward(cp, nf, ef, adj, src, neg, t, P, L)
            pos, ng = orc.merge_layer(cm, ps, pd).squeeze(-1).sigmoid(), orc.merge_layer(cm, ns, nd).squeeze(-1).sigmoid()
            closs = torch.nn.functional.binary_cross_entropy(torch.cat([pos, ng]), torch.cat([torch.ones_like(pos), torch.zeros_like(ng)]))
            copt.zero_grad()
            closs.backward()
            copt.step()
        ncpu, cel, _ = _timed_cpu(cstep, 0, min(cpu_max_steps, steps + warmup - 1), cpu_budget_s, warm=1)
        out["cpu_baseline"] = {"value": round(ncpu * B / cel, 1), "unit": "edges/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{ncpu} of the same steps ({cel:.1f} s) after 1 warm-up step, torch autograd through oracle/dygformer_oracle.py"}
    return out


SECONDARY = {"lastfm": bench_lastfm, "tgat": bench_tgat, "tgn": bench_tgn, "train": bench_train}


# ======================================================================================================================
def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="wikipedia", choices=list(WORKLOADS))
    ap.add_argument("--impl", type=int, default=0, help="0 auto, 1 generic kernels, 3 fused kernel (token-owner)")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the launches are issued on (round-robin)")
    ap.add_argument("--fuse-steps", type=int, default=32,
                    help="upper bound of the steps per launch: the positive and negative calls of F consecutive steps (2F independently "
                         "padded groups of `batch` pairs) form ONE grid; F = min(this, steps/2) so that at least two launches are timed")
    ap.add_argument("--prime-launches", type=int, default=12, help="untimed launches of the timed shape before the warm-up steps (clock / cache ramp)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall budget of the headline CPU-baseline sample (0 = skip CPU legs and parity)")
    ap.add_argument("--secondary", default="lastfm,tgat,tgn,train", help="comma list of secondary workloads to run at N=1 ('' or 'none' = skip)")
    ap.add_argument("--full-span-oracle", action="store_true", help="stages.full_span: recompute ALL 237 batches' AP / AUC with the CPU oracle (~1 min)")
    ap.add_argument("--sharded-secondary", default="lastfm", choices=["lastfm", "tiny", "none"],
                    help="N > 1: workload that is also run sharded over the N ranks and printed as secondary.<name> (BASELINE config 4 = lastfm)")
    ap.add_argument("--budget-seconds", type=float, default=105.0, help="wall budget of the whole run: secondary CPU samples shrink to fit")
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
    # One process per GPU over RCCL ("nccl").  Rehearsal of the N > 1 path on a box with fewer GPUs than ranks (tests, the builder's
    # one-GPU box): DYGNN_BENCH_BACKEND=gloo shares the devices round-robin and reduces through gloo — same code path, no xGMI.
    backend = os.environ.get("DYGNN_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if world > 1 and backend == "nccl" and local_rank >= n_dev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {n_dev} GPU(s) visible (RCCL needs one GPU per rank)")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, n_dev)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    # DYGNN_BENCH_FORCE_DIST=1: the process group (and with it the barrier, the MAX-reduce of the timed region and the float64 metric
    # all-reduce on device tensors) is set up at world size 1 too — RCCL's first contact on a one-GPU box (tests/test_bench_multirank_gpu.py)
    force_dist = bool(os.environ.get("DYGNN_BENCH_FORCE_DIST")) and world == 1
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    # one node: the graph is generated once (local rank 0) and mapped by the other ranks
    share = (dist, local_rank, f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}") if world > 1 else None

    wk = DygformerWorkload(args.workload, dev, args.impl, share=share)
    B, L, P = wk.B, wk.L, wk.P
    F = max(1, min(args.fuse_steps, args.steps // 2 if args.steps >= 2 else 1))
    solo = rank == 0 and world == 1
    keep = min(args.steps, 64) if solo and args.cpu_seconds > 0 else 0
    res = run_dygformer(wk, args.steps, args.warmup, F, rank, world, dist, args.streams, keep=keep, prime=args.prime_launches)

    out = {
        "metric": "edges/sec (link-prediction fwd) DyGFormer " + WORKLOAD_LABEL[args.workload],
        "value": round(res["value"], 1), "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(res["elapsed"] / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": wk.describe(), "batch": B, "max_input_sequence_length": L, "patch_size": P,
                   "parallelism": f"{world} x edge-batch shard, graph+weights replicated, metric all-reduce over " + ("RCCL" if backend == "nccl" else backend),
                   "impl": {0: "auto", 1: "generic", 3: "fused3"}.get(args.impl, str(args.impl)), "streams": res["streams"],
                   "steps_per_launch": res["steps_per_launch"], "launches_timed": res["n_launch"],
                   "untimed": f"{max(1, args.prime_launches)} launches of the timed shape + {args.warmup} warm-up steps"},
        "roofline": dygformer_roofline(wk, res, args.impl),
        "mean_auc": round(res["mean_auc"], 4),
    }
    if dist is not None:
        # the one exchange of the sharded evaluation, after the timed region: every rank's [sum AUC, sum gap, #steps] summed over the backend
        out["metric_allreduce"] = {"backend": "RCCL (nccl)" if backend == "nccl" else backend, "ms": round(res["metric_allreduce_ms"], 3),
                                   "steps_counted": res["metric_steps"], "steps_expected": args.steps * world}
        if res["metric_steps"] != args.steps * world:
            raise SystemExit(f"metric all-reduce counted {res['metric_steps']} steps, expected {args.steps * world}")
    failed = False
    if world > 1 and args.sharded_secondary != "none":
        # BASELINE config 4 is defined at N GPUs: the LastFM-shaped workload (L=512, P=8) sharded exactly like the headline, every rank K' steps
        del wk
        torch.cuda.empty_cache()
        wk2 = DygformerWorkload(args.sharded_secondary, dev, args.impl, share=share)
        st2, wu2, F2 = 16, 8, 8
        res2 = run_dygformer(wk2, st2, wu2, F2, rank, world, dist, 1, keep=0, prime=4)
        out["secondary"] = {args.sharded_secondary: {
            "metric": f"edges/sec (link-prediction fwd) DyGFormer {WORKLOAD_LABEL[args.sharded_secondary]}, {world} ranks, edge-batch shard",
            "value": round(res2["value"], 1), "unit": "edges/s", "n_gpus": world, "steps": st2, "warmup": wu2, "scaling": "weak",
            "ms_per_step": round(res2["elapsed"] / st2 * 1e3, 4), "steps_per_launch": res2["steps_per_launch"],
            "config": {"workload": wk2.describe()}, "roofline": dygformer_roofline(wk2, res2, args.impl), "mean_auc": round(res2["mean_auc"], 4),
            "metric_allreduce": {"ms": round(res2["metric_allreduce_ms"], 3), "steps_counted": res2["metric_steps"]}}}
        del wk2
    if solo:
        out["stages"] = sampler_stage(wk.sampler, wk.data, dev)           # stage-level number for the neighbour lookup
        out["stages"].update(metrics_stage(dev))
        out["stages"].update(per_call_stage(wk))
        out["stages"].update(full_span_stage(wk, oracle_batches=6 if args.cpu_seconds > 0 else 0, oracle_all=args.full_span_oracle))
        failed |= not out["stages"]["full_span"].get("oracle_check", {"ok": True})["ok"]
        if args.cpu_seconds > 0:
            out["cpu_baseline"], out["parity"] = cpu_baseline_and_parity(wk, res, args.cpu_seconds, 32, one_thread_s=0.4 * args.cpu_seconds)
            out["speedup_vs_cpu_baseline"] = round(res["value"] / out["cpu_baseline"]["value"], 1)
            failed |= not out["parity"]["ok"]
        res["kept"] = None
        names = [s for s in args.secondary.split(",") if s and s != "none"]
        if names:
            del wk
            torch.cuda.empty_cache()
            out["secondary"] = {}
            # CPU budgets: what is left of the run's wall budget, shared by the legs still to run (GPU parts take ~5 s each)
            nominal = {"lastfm": 10.0, "tgat": 20.0, "tgn": 8.0, "train": 12.0}
            for i, name in enumerate(names):
                left = args.budget_seconds - (time.perf_counter() - T_START) - 6.0 * (len(names) - i)
                share = max(0.0, left) * nominal[name] / sum(nominal[n] for n in names[i:])
                kw = {"cpu_budget_s": 0.0 if args.cpu_seconds <= 0 else max(1.5, min(nominal[name], share))}
                try:
                    t_leg = time.perf_counter()
                    r = SECONDARY[name](dev, **kw)
                    r["wall_s"] = round(time.perf_counter() - t_leg, 1)
                    out["secondary"][name] = r
                    if "parity" in r:
                        failed |= not r["parity"]["ok"]
                except Exception as e:                      # a secondary leg never takes the headline line down; it is reported
                    out["secondary"][name] = {"error": f"{type(e).__name__}: {e}"}
                    failed = True
                torch.cuda.empty_cache()
        out["wall_s"] = round(time.perf_counter() - T_START, 1)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if failed:
        raise SystemExit("bench.py: parity above tolerance or a secondary workload failed (see the JSON line)")


if __name__ == "__main__":
    main()
