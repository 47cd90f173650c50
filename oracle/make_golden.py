#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE itself (imported from /root/reference) on the
seeded recipes of tests/golden_cases.py.  Run in the build container only:

    python oracle/make_golden.py            # writes tests/golden/<case>.npz

The reference never travels: only its outputs are stored (inputs are rebuilt from the recipe).
Test infrastructure — see oracle/dygformer_oracle.py header.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DYGLIB_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

from models.DyGFormer import DyGFormer as RefDyGFormer          # noqa: E402  (reference)
from models.modules import MergeLayer as RefMergeLayer          # noqa: E402  (reference)
from models.TGAT import TGAT as RefTGAT                         # noqa: E402  (reference)
from models.MemoryModel import MemoryModel as RefMemoryModel    # noqa: E402  (reference)
from utils.DataLoader import Data as RefData                    # noqa: E402  (reference)
from utils.utils import get_neighbor_sampler as ref_get_neighbor_sampler  # noqa: E402  (reference)

from tests import golden_cases as gc                             # noqa: E402


def run_case(name: str) -> dict:
    c = gc.build_case(name)
    d, cfg = c["data"], c["cfg"]
    ref_data = RefData(d.src_node_ids, d.dst_node_ids, d.node_interact_times, d.edge_ids, d.labels)
    sampler = ref_get_neighbor_sampler(ref_data, sample_neighbor_strategy="recent", seed=1)
    model = RefDyGFormer(c["node_feat"], c["edge_feat"], sampler, time_feat_dim=cfg["time_feat_dim"],
                         channel_embedding_dim=cfg["channel_embedding_dim"], patch_size=cfg["patch_size"],
                         num_layers=cfg["num_layers"], num_heads=cfg["num_heads"], dropout=0.1,
                         max_input_sequence_length=cfg["max_input_sequence_length"], device="cpu")
    missing = model.load_state_dict({k: torch.from_numpy(v) for k, v in c["params"].items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    merge = RefMergeLayer(172, 172, 172, 1)
    merge.load_state_dict({k: torch.from_numpy(v) for k, v in c["mparams"].items()}, strict=True)
    model.eval(), merge.eval()

    out = {}
    src, dst, neg, t = c["src"], c["dst"], c["neg_dst"], c["times"]

    # --- sampler: get_historical_neighbors('recent') on src and dst queries
    q_nodes = np.concatenate([src, dst])
    q_times = np.concatenate([t, t])
    for k in gc.SAMPLER_KS:
        n, e, ts = sampler.get_historical_neighbors(q_nodes, q_times, num_neighbors=k)
        out[f"recent{k}_nbr"], out[f"recent{k}_eid"], out[f"recent{k}_ts"] = n, e, ts
    # TGAT-style second hop: float32-rounded query times (models/TGAT.py:107-110)
    n, e, ts = sampler.get_historical_neighbors(q_nodes, q_times, num_neighbors=10)
    n2, e2, ts2 = sampler.get_historical_neighbors(n.flatten(), ts.flatten(), num_neighbors=10)
    out["hop2_nbr"], out["hop2_eid"], out["hop2_ts"] = n2, e2, ts2

    # --- first-hop counts, padded windows and co-occurrence counts for (src, dst)
    sides = {}
    for tag, ids in (("src", src), ("dst", dst)):
        a, b, cc = sampler.get_all_first_hop_neighbors(ids, t)
        out[f"{tag}_hist_len"] = np.array([len(x) for x in a], dtype=np.int64)
        pid, pe, pt = model.pad_sequences(ids, t, a, b, cc, patch_size=cfg["patch_size"],
                                          max_input_sequence_length=cfg["max_input_sequence_length"])
        sides[tag] = pid
        out[f"{tag}_pad_ids"], out[f"{tag}_pad_eids"], out[f"{tag}_pad_times"] = pid, pe, pt
    cs, cd = model.neighbor_co_occurrence_encoder.count_nodes_appearances(sides["src"], sides["dst"])
    out["src_counts"], out["dst_counts"] = cs.numpy(), cd.numpy()

    # --- full forward with taps
    taps = {"layers": []}
    h0 = model.transformers[0].register_forward_pre_hook(lambda m, inp: taps.__setitem__("enc_in", inp[0].detach().clone()))
    hs = [tr.register_forward_hook(lambda m, inp, o: taps["layers"].append(o.detach().clone())) for tr in model.transformers]
    with torch.no_grad():
        se, de = model.compute_src_dst_node_temporal_embeddings(src, dst, t)
    h0.remove()
    [h.remove() for h in hs]
    R = gc.TAP_ROWS
    out["encoder_input_rows"] = taps["enc_in"][:R].numpy()
    for l, x in enumerate(taps["layers"]):
        out[f"layer{l}_rows"] = x[:R].numpy()
    out["src_emb"], out["dst_emb"] = se.numpy(), de.numpy()
    with torch.no_grad():
        nse, nde = model.compute_src_dst_node_temporal_embeddings(src, neg, t)
        out["neg_src_emb"], out["neg_dst_emb"] = nse.numpy(), nde.numpy()
        out["pos_prob"] = merge(se, de).squeeze(-1).sigmoid().numpy()
        out["neg_prob"] = merge(nse, nde).squeeze(-1).sigmoid().numpy()
    out["torch_version"] = np.array(torch.__version__)
    return out


def run_sampling_case(name: str) -> dict:
    """uniform / time_interval_aware draws of the reference sampler (utils/utils.py:176-199) on the case's query batch."""
    c = gc.build_case(name)
    d = c["data"]
    ref_data = RefData(d.src_node_ids, d.dst_node_ids, d.node_interact_times, d.edge_ids, d.labels)
    q_nodes = np.concatenate([c["src"], c["dst"]])
    q_times = np.concatenate([c["times"], c["times"]])
    out = {}
    for tag, (strategy, seed, tsf) in gc.SAMPLING_STRATEGIES.items():
        sampler = ref_get_neighbor_sampler(ref_data, sample_neighbor_strategy=strategy, time_scaling_factor=tsf, seed=seed)
        for k in gc.SAMPLING_KS:
            n, e, t = sampler.get_historical_neighbors(q_nodes, q_times, num_neighbors=k)
            out[f"{tag}_k{k}_nbr"], out[f"{tag}_k{k}_eid"], out[f"{tag}_k{k}_ts"] = n, e, t
        nl, el, tl = sampler.get_multi_hop_neighbors(2, q_nodes, q_times, num_neighbors=gc.SAMPLING_HOP_K)
        for h in range(2):
            out[f"{tag}_hop{h}_nbr"], out[f"{tag}_hop{h}_eid"], out[f"{tag}_hop{h}_ts"] = nl[h], el[h], tl[h]
        if strategy == "time_interval_aware":       # probabilities of the busiest node (compute_sampled_probabilities)
            v = int(np.argmax([len(x) for x in sampler.nodes_neighbor_ids]))
            out[f"{tag}_prob_node"] = np.array(v)
            out[f"{tag}_prob"] = np.asarray(sampler.nodes_neighbor_sampled_probabilities[v], dtype=np.float64)
    out["numpy_version"] = np.array(np.__version__)
    return out


def run_loader_case() -> dict:
    """the reference's get_link_prediction_data on the synthetic dataset files of tests/golden_cases.py"""
    import contextlib, io, tempfile, warnings
    from utils.DataLoader import get_link_prediction_data as ref_get_link_prediction_data      # noqa: E402  (reference)
    tmp = tempfile.mkdtemp()
    name = gc.write_dataset_files(tmp)
    cwd = os.getcwd()
    os.chdir(tmp)                     # the reference reads ./processed_data/...
    try:
        with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = ref_get_link_prediction_data(name, gc.LOADER_CASE["val_ratio"], gc.LOADER_CASE["test_ratio"])
    finally:
        os.chdir(cwd)
    out = {"node_feat_shape": np.array(r[0].shape), "edge_feat_shape": np.array(r[1].shape),
           "node_feat_dtype": np.array(str(r[0].dtype)), "edge_feat_dtype": np.array(str(r[1].dtype)),
           "edge_feat_sum": np.array(r[1].sum()), "python_version": np.array(sys.version.split()[0])}
    for tag, d in zip(("full", "train", "val", "test", "new_node_val", "new_node_test"), r[2:]):
        out[f"{tag}_edge_ids"] = d.edge_ids
        out[f"{tag}_num_unique_nodes"] = np.array(d.num_unique_nodes)
    return out


def run_grad_case(name: str) -> dict:
    """parameter gradients of the reference DyGFormer (eval mode: dropout identity, autograd on)"""
    c = gc.build_case(name)
    d, cfg = c["data"], c["cfg"]
    ref_data = RefData(d.src_node_ids, d.dst_node_ids, d.node_interact_times, d.edge_ids, d.labels)
    sampler = ref_get_neighbor_sampler(ref_data, sample_neighbor_strategy="recent", seed=1)
    model = RefDyGFormer(c["node_feat"], c["edge_feat"], sampler, time_feat_dim=cfg["time_feat_dim"],
                         channel_embedding_dim=cfg["channel_embedding_dim"], patch_size=cfg["patch_size"],
                         num_layers=cfg["num_layers"], num_heads=cfg["num_heads"], dropout=0.1,
                         max_input_sequence_length=cfg["max_input_sequence_length"], device="cpu")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in c["params"].items()}, strict=True)
    model.eval()
    G1, G2 = gc.grad_loss_weights(len(c["src"]))
    se, de = model.compute_src_dst_node_temporal_embeddings(c["src"], c["dst"], c["times"])
    loss = (se * torch.from_numpy(G1)).sum() + (de * torch.from_numpy(G2)).sum()
    loss.backward()
    out = {"loss": np.array(float(loss.detach()))}
    for k, p in model.named_parameters():
        out.update(gc.grad_signature(k, p.grad.numpy()))
    return out


def run_tgat_case(name: str) -> dict:
    c = gc.build_tgat_case(name)
    d, cfg = c["data"], c["tgat_cfg"]
    ref_data = RefData(d.src_node_ids, d.dst_node_ids, d.node_interact_times, d.edge_ids, d.labels)
    sampler = ref_get_neighbor_sampler(ref_data, sample_neighbor_strategy="recent", seed=1)
    model = RefTGAT(c["node_feat"], c["edge_feat"], sampler, time_feat_dim=cfg["time_feat_dim"], num_layers=cfg["num_layers"],
                    num_heads=cfg["num_heads"], dropout=0.1, device="cpu")
    r = model.load_state_dict({k: torch.from_numpy(v) for k, v in c["tgat_params"].items()}, strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    model.eval()
    with torch.no_grad():
        se, de = model.compute_src_dst_node_temporal_embeddings(c["src"], c["dst"], c["times"], num_neighbors=cfg["num_neighbors"])
        nse, nde = model.compute_src_dst_node_temporal_embeddings(c["src"], c["neg_dst"], c["times"], num_neighbors=cfg["num_neighbors"])
    return {"src_emb": se.numpy(), "dst_emb": de.numpy(), "neg_src_emb": nse.numpy(), "neg_dst_emb": nde.numpy(),
            "torch_version": np.array(torch.__version__)}


def run_tgat_random_case(name: str) -> dict:
    c = gc.build_tgat_case(name)
    d, cfg = c["data"], c["tgat_cfg"]
    ref_data = RefData(d.src_node_ids, d.dst_node_ids, d.node_interact_times, d.edge_ids, d.labels)
    out = {}
    for tag, (strategy, seed, tsf) in gc.SAMPLING_STRATEGIES.items():
        sampler = ref_get_neighbor_sampler(ref_data, sample_neighbor_strategy=strategy, time_scaling_factor=tsf, seed=seed)
        model = RefTGAT(c["node_feat"], c["edge_feat"], sampler, time_feat_dim=cfg["time_feat_dim"], num_layers=cfg["num_layers"],
                        num_heads=cfg["num_heads"], dropout=0.1, device="cpu")
        model.load_state_dict({k: torch.from_numpy(v) for k, v in c["tgat_params"].items()}, strict=True)
        model.eval()
        with torch.no_grad():
            se, de = model.compute_src_dst_node_temporal_embeddings(c["src"], c["dst"], c["times"], num_neighbors=cfg["num_neighbors"])
            nse, nde = model.compute_src_dst_node_temporal_embeddings(c["src"], c["neg_dst"], c["times"], num_neighbors=cfg["num_neighbors"])
        out[f"{tag}_src_emb"], out[f"{tag}_dst_emb"], out[f"{tag}_neg_src_emb"], out[f"{tag}_neg_dst_emb"] = se.numpy(), de.numpy(), nse.numpy(), nde.numpy()
    return out


def run_tgn_case(name: str) -> dict:
    c = gc.build_tgn_case(name)
    d, cfg = c["data"], c["tgn_cfg"]
    # the sampler must only know interactions... the reference evaluates with the full graph sampler; so do we
    ref_data = RefData(d.src_node_ids, d.dst_node_ids, d.node_interact_times, d.edge_ids, d.labels)
    sampler = ref_get_neighbor_sampler(ref_data, sample_neighbor_strategy="recent", seed=1)
    model = RefMemoryModel(c["node_feat"], c["edge_feat"], sampler, time_feat_dim=cfg["time_feat_dim"], model_name="TGN",
                           num_layers=cfg["num_layers"], num_heads=cfg["num_heads"], dropout=0.1, device="cpu")
    sd = model.state_dict()
    for k, v in c["tgn_params"].items():
        assert k in sd and tuple(sd[k].shape) == v.shape, k
        sd[k] = torch.from_numpy(v)
    model.load_state_dict(sd, strict=True)
    model.eval()
    model.memory_bank.__init_memory_bank__()
    out = {}
    with torch.no_grad():
        for i, b in enumerate(c["tgn_batches"]):
            ns, nd = model.compute_src_dst_node_temporal_embeddings(b["src"], b["neg"], b["t"], edge_ids=None, edges_are_positive=False,
                                                                    num_neighbors=cfg["num_neighbors"])
            ps, pd = model.compute_src_dst_node_temporal_embeddings(b["src"], b["dst"], b["t"], edge_ids=b["eid"], edges_are_positive=True,
                                                                    num_neighbors=cfg["num_neighbors"])
            out[f"b{i}_neg_src"], out[f"b{i}_neg_dst"], out[f"b{i}_pos_src"], out[f"b{i}_pos_dst"] = ns.numpy(), nd.numpy(), ps.numpy(), pd.numpy()
    out["final_memory"] = model.memory_bank.node_memories.data.numpy().copy()
    out["final_last_update"] = model.memory_bank.node_last_updated_times.data.numpy().copy()
    out["state_dict_keys"] = np.array(sorted(model.state_dict().keys()))
    out["torch_version"] = np.array(torch.__version__)
    return out


def run_tgn_random_case(name: str) -> dict:
    """run_tgn_case with the random sampling strategies (MemoryModel.py:626-629 calls whatever sampler it holds)"""
    c = gc.build_tgn_case(name)
    d, cfg = c["data"], c["tgn_cfg"]
    ref_data = RefData(d.src_node_ids, d.dst_node_ids, d.node_interact_times, d.edge_ids, d.labels)
    out = {}
    for tag, (strategy, seed, tsf) in gc.SAMPLING_STRATEGIES.items():
        sampler = ref_get_neighbor_sampler(ref_data, sample_neighbor_strategy=strategy, time_scaling_factor=tsf, seed=seed)
        model = RefMemoryModel(c["node_feat"], c["edge_feat"], sampler, time_feat_dim=cfg["time_feat_dim"], model_name="TGN",
                               num_layers=cfg["num_layers"], num_heads=cfg["num_heads"], dropout=0.1, device="cpu")
        sd = model.state_dict()
        for k, v in c["tgn_params"].items():
            sd[k] = torch.from_numpy(v)
        model.load_state_dict(sd, strict=True)
        model.eval()
        model.memory_bank.__init_memory_bank__()
        with torch.no_grad():
            for i, b in enumerate(c["tgn_batches"]):
                ns, nd = model.compute_src_dst_node_temporal_embeddings(b["src"], b["neg"], b["t"], edge_ids=None, edges_are_positive=False,
                                                                        num_neighbors=cfg["num_neighbors"])
                ps, pd = model.compute_src_dst_node_temporal_embeddings(b["src"], b["dst"], b["t"], edge_ids=b["eid"], edges_are_positive=True,
                                                                        num_neighbors=cfg["num_neighbors"])
                out[f"{tag}_b{i}_neg_src"], out[f"{tag}_b{i}_neg_dst"], out[f"{tag}_b{i}_pos_src"], out[f"{tag}_b{i}_pos_dst"] = ns.numpy(), nd.numpy(), ps.numpy(), pd.numpy()
        out[f"{tag}_final_memory"] = model.memory_bank.node_memories.data.numpy().copy()
        out[f"{tag}_final_last_update"] = model.memory_bank.node_last_updated_times.data.numpy().copy()
    return out


def run_eval_case(name: str) -> dict:
    """the reference's evaluate_model_link_prediction (evaluate_models_utils.py:18-153) end to end on the CPU"""
    import contextlib, io
    from evaluate_models_utils import evaluate_model_link_prediction as ref_evaluate          # noqa: E402  (reference)
    from utils.utils import NegativeEdgeSampler as RefNegativeEdgeSampler                      # noqa: E402  (reference)
    from utils.DataLoader import get_idx_data_loader as ref_get_idx_data_loader               # noqa: E402  (reference)
    r = gc.EVAL_CASES[name]
    if r["model"] == "DyGFormer":
        c = gc.build_case(r["graph"]); cfg = c["cfg"]
    elif r["model"] == "TGAT":
        c = gc.build_tgat_case(r["graph"]); cfg = c["tgat_cfg"]
    else:
        c = gc.build_tgn_case(r["graph"]); cfg = c["tgn_cfg"]
    d = c["data"]
    ref_data = RefData(d.src_node_ids, d.dst_node_ids, d.node_interact_times, d.edge_ids, d.labels)
    sampler = ref_get_neighbor_sampler(ref_data, sample_neighbor_strategy=r.get("strategy", "recent"), seed=r.get("sampler_seed", 1))
    if r["model"] == "DyGFormer":
        backbone = RefDyGFormer(c["node_feat"], c["edge_feat"], sampler, time_feat_dim=cfg["time_feat_dim"],
                                channel_embedding_dim=cfg["channel_embedding_dim"], patch_size=cfg["patch_size"], num_layers=cfg["num_layers"],
                                num_heads=cfg["num_heads"], dropout=0.1, max_input_sequence_length=cfg["max_input_sequence_length"], device="cpu")
        backbone.load_state_dict({k: torch.from_numpy(v) for k, v in c["params"].items()}, strict=True)
    elif r["model"] == "TGAT":
        backbone = RefTGAT(c["node_feat"], c["edge_feat"], sampler, time_feat_dim=cfg["time_feat_dim"], num_layers=cfg["num_layers"],
                           num_heads=cfg["num_heads"], dropout=0.1, device="cpu")
        backbone.load_state_dict({k: torch.from_numpy(v) for k, v in c["tgat_params"].items()}, strict=True)
    else:
        backbone = RefMemoryModel(c["node_feat"], c["edge_feat"], sampler, time_feat_dim=cfg["time_feat_dim"], model_name="TGN",
                                  num_layers=cfg["num_layers"], num_heads=cfg["num_heads"], dropout=0.1, device="cpu")
        sd = backbone.state_dict(); sd.update({k: torch.from_numpy(v) for k, v in c["tgn_params"].items()}); backbone.load_state_dict(sd, strict=True)
        backbone.memory_bank.__init_memory_bank__()
    merge = RefMergeLayer(172, 172, 172, 1)
    merge.load_state_dict({k: torch.from_numpy(v) for k, v in c["mparams"].items()}, strict=True)
    model = torch.nn.Sequential(backbone, merge)
    first, last = gc.eval_indices(d.num_interactions)
    sl = slice(first, last)
    eval_data = RefData(d.src_node_ids[sl], d.dst_node_ids[sl], d.node_interact_times[sl], d.edge_ids[sl], d.labels[sl])
    neg = RefNegativeEdgeSampler(src_node_ids=d.src_node_ids, dst_node_ids=d.dst_node_ids, seed=gc.EVAL_NEG_SEED)
    loader = ref_get_idx_data_loader(indices_list=list(range(last - first)), batch_size=r["batch"], shuffle=False)
    with contextlib.redirect_stderr(io.StringIO()):
        losses, metrics = ref_evaluate(model_name=r["model"], model=model, neighbor_sampler=sampler, evaluate_idx_data_loader=loader,
                                       evaluate_neg_edge_sampler=neg, evaluate_data=eval_data, loss_func=torch.nn.BCELoss(),
                                       num_neighbors=cfg.get("num_neighbors", 20), time_gap=2000)
    neg.reset_random_state()
    draws = [neg.sample(size=min(r["batch"], last - first - i))[1] for i in range(0, last - first, r["batch"])]
    return {"losses": np.array(losses, dtype=np.float64), "average_precision": np.array([m["average_precision"] for m in metrics]),
            "roc_auc": np.array([m["roc_auc"] for m in metrics]), "neg_dst": np.concatenate(draws),
            "neg_src_first": neg.random_state.randint(0, 10, 1)}          # state check: the next draw after the whole pass


def run_metric_cases() -> dict:
    """the reference's utils/metrics.py (scikit-learn underneath) and torch's BCELoss (train_link_prediction.py:147 loss_func)
    on the score vectors of tests/golden_cases.py"""
    import sklearn
    from utils.metrics import get_link_prediction_metrics as ref_link, get_node_classification_metrics as ref_node   # noqa: E402  (reference)
    out = {"sklearn_version": np.array(sklearn.__version__)}
    for name in gc.METRIC_CASES:
        p, y = gc.build_metric_case(name)
        pt, yt = torch.from_numpy(p), torch.from_numpy(y)
        m = ref_link(predicts=pt, labels=yt)
        out[f"{name}|average_precision"] = np.array(m["average_precision"], dtype=np.float64)
        out[f"{name}|roc_auc"] = np.array(m["roc_auc"], dtype=np.float64)
        out[f"{name}|node_roc_auc"] = np.array(ref_node(predicts=pt, labels=yt)["roc_auc"], dtype=np.float64)
        out[f"{name}|bce_loss"] = np.array(torch.nn.BCELoss()(input=pt, target=yt).item(), dtype=np.float64)
    return out


def main():
    os.makedirs(gc.GOLDEN_DIR, exist_ok=True)
    torch.set_num_threads(8)
    names = sys.argv[1:] or (list(gc.CASES) + list(gc.TGAT_CASES) + list(gc.TGN_CASES) + ["sampling_" + n for n in gc.SAMPLING_CASES] + ["loader_toy"] + ["grads_" + n for n in gc.GRAD_CASES] + ["tgat_rand_" + n for n in gc.TGAT_RANDOM_CASES] + ["tgn_rand_" + n for n in gc.TGN_RANDOM_CASES] + ["metrics"] + list(gc.EVAL_CASES))
    for name in names:
        if name in gc.EVAL_CASES:
            np.savez_compressed(os.path.join(gc.GOLDEN_DIR, name + ".npz"), **run_eval_case(name))
            print(f"{name}: written")
            continue
        if name == "metrics":
            np.savez_compressed(os.path.join(gc.GOLDEN_DIR, "metrics.npz"), **run_metric_cases())
            print("metrics: written")
            continue
        if name.startswith("tgn_rand_"):
            path = os.path.join(gc.GOLDEN_DIR, name + ".npz")
            np.savez_compressed(path, **run_tgn_random_case(name[len("tgn_rand_"):]))
            print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")
            continue
        if name.startswith("tgat_rand_"):
            path = os.path.join(gc.GOLDEN_DIR, name + ".npz")
            np.savez_compressed(path, **run_tgat_random_case(name[len("tgat_rand_"):]))
            print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")
            continue
        if name.startswith("grads_"):
            path = os.path.join(gc.GOLDEN_DIR, name + ".npz")
            np.savez_compressed(path, **run_grad_case(name[len("grads_"):]))
            print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")
            continue
        if name == "loader_toy":
            np.savez_compressed(os.path.join(gc.GOLDEN_DIR, name + ".npz"), **run_loader_case())
            print(f"{name}: written")
            continue
        if name.startswith("sampling_"):
            out = run_sampling_case(name[len("sampling_"):])
            path = os.path.join(gc.GOLDEN_DIR, name + ".npz")
            np.savez_compressed(path, **out)
            print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")
            continue
        if name in gc.TGN_CASES:
            out = run_tgn_case(name)
            path = os.path.join(gc.GOLDEN_DIR, name + ".npz")
            np.savez_compressed(path, **out)
            print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")
            continue
        if name in gc.TGAT_CASES:
            out = run_tgat_case(name)
            path = os.path.join(gc.GOLDEN_DIR, name + ".npz")
            np.savez_compressed(path, **out)
            print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB  B={len(out['src_emb'])}")
            continue
        out = run_case(name)
        path = os.path.join(gc.GOLDEN_DIR, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB  "
              f"S_src={out['src_pad_ids'].shape[1]} S_dst={out['dst_pad_ids'].shape[1]} B={len(out['src_emb'])}")


if __name__ == "__main__":
    main()
