#!/usr/bin/env python3
"""End-to-end miniature of the reference's train_link_prediction.py on the HIP path, with synthetic dataset files in the
reference's on-disk format: load (`get_link_prediction_data`), build the two samplers (train graph / full graph,
train_link_prediction.py:40-45), train DyGFormer + MergeLayer with Adam on BCE (:229-257), evaluate AP / AUC on the
validation split with the fused inference kernel (evaluate_models_utils.py:49-152).  One process per GPU under
torch.distributed.run averages gradients with one flat RCCL all-reduce per step.

    python examples/train_link_prediction_synthetic.py --epochs 2
"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dyglib_amd import DyGFormer, MergeLayer, get_link_prediction_data, get_neighbor_sampler, synthetic as syn  # noqa: E402
from dyglib_amd import distributed as D  # noqa: E402


def write_files(root, name, users, items, edges, seed):
    d = os.path.join(root, name)
    os.makedirs(d, exist_ok=True)
    data, nf, ef = syn.make_bipartite_graph(users, items, edges, seed=seed)
    with open(os.path.join(d, f"ml_{name}.csv"), "w") as f:
        f.write(",u,i,ts,label,idx\n")
        for k in range(data.num_interactions):
            f.write(f"{k},{int(data.src_node_ids[k])},{int(data.dst_node_ids[k])},{float(data.node_interact_times[k])!r},0.0,{int(data.edge_ids[k])}\n")
    np.save(os.path.join(d, f"ml_{name}.npy"), ef)
    np.save(os.path.join(d, f"ml_{name}_node.npy"), nf)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--batch", type=int, default=200)
    ap.add_argument("--users", type=int, default=400)
    ap.add_argument("--items", type=int, default=60)
    ap.add_argument("--edges", type=int, default=20000)
    ap.add_argument("--lr", type=float, default=1e-4)
    args = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        dist.init_process_group("nccl")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)

    root = tempfile.mkdtemp()
    write_files(root, "toy", args.users, args.items, args.edges, seed=0)
    node_feat, edge_feat, full, train, val, test, nn_val, nn_test = get_link_prediction_data("toy", 0.15, 0.15, root=root)
    train_sampler = get_neighbor_sampler(train, "recent", seed=0, device=dev)
    full_sampler = get_neighbor_sampler(full, "recent", seed=1, device=dev)
    model = DyGFormer(node_feat, edge_feat, train_sampler, time_feat_dim=100, channel_embedding_dim=50, patch_size=1, num_layers=2,
                      num_heads=2, dropout=0.1, max_input_sequence_length=32, device=dev).to(dev)
    merge = MergeLayer(172, 172, 172, 1).to(dev)
    params = list(model.parameters()) + list(merge.parameters())
    opt = torch.optim.Adam(params, lr=args.lr)
    rs = np.random.RandomState(0)
    items = np.unique(full.dst_node_ids)
    train_items = np.unique(train.dst_node_ids)      # training negatives come from the TRAINING graph's destinations (train_link_prediction.py:95-96)

    def evaluate(split):
        model.eval(); merge.eval()
        model.set_neighbor_sampler(full_sampler)
        nb = (split.num_interactions + args.batch - 1) // args.batch
        ers = np.random.RandomState(1)

        def step(i):
            sl = slice(i * args.batch, (i + 1) * args.batch)
            src, dst, t = split.src_node_ids[sl], split.dst_node_ids[sl], split.node_interact_times[sl]
            neg = ers.choice(items, size=len(src))
            with torch.no_grad():
                es, ed = model.compute_src_dst_node_temporal_embeddings_many(np.stack([src, src]), np.stack([dst, neg]), np.stack([t, t]))
                prob = merge.link_probabilities(es.flatten(0, 1), ed.flatten(0, 1))
            return prob[:len(src)], prob[len(src):]
        return D.evaluate_sharded(step, nb, rank, world, device=dev)

    history = []
    for epoch in range(args.epochs):
        model.train(); merge.train()
        model.set_neighbor_sampler(train_sampler)
        nb = train.num_interactions // args.batch

        def train_step(i):
            sl = slice(i * args.batch, (i + 1) * args.batch)
            src, dst, t = train.src_node_ids[sl], train.dst_node_ids[sl], train.node_interact_times[sl]
            neg = rs.choice(train_items, size=len(src))
            # the positive and the negative call of the step (train_link_prediction.py:229-239) as one set: one dense pass when they pad alike
            s2, d2 = model.compute_src_dst_node_temporal_embeddings_many(np.stack([src, src]), np.stack([dst, neg]), np.stack([t, t]))
            ps, pd, ns, nd = s2[0], d2[0], s2[1], d2[1]
            pos, ng = merge(ps, pd).squeeze(-1).sigmoid(), merge(ns, nd).squeeze(-1).sigmoid()
            return torch.nn.functional.binary_cross_entropy(torch.cat([pos, ng]), torch.cat([torch.ones_like(pos), torch.zeros_like(ng)]))
        # every rank takes ceil(nb / world) optimizer steps; a rank without a batch joins the gradient all-reduce with zeros
        losses = D.train_sharded(train_step, params, opt, nb, rank, world) or [float("nan")]
        m = evaluate(val)
        history.append({"epoch": epoch, "train_loss": float(np.mean(losses)), "val_ap": m["average_precision"], "val_auc": m["roc_auc"]})
        if rank == 0:
            print(json.dumps(history[-1]), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return history


if __name__ == "__main__":
    main()
