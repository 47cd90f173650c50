"""N>1 path on CPU: world_size-2 gloo processes run the sharded evaluation loop and the sharded training loop of
dyglib_amd/distributed.py (the code bench.py and the examples run on RCCL); the reduced metrics must equal the
single-process result and an uneven batch split must not hang.  The per-batch metrics come from the device kernel on a
GPU; on CPU the loop is handed the oracle's estimators (oracle/metrics_oracle.py, test-only) through `metrics_fn`."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dyglib_amd import distributed as D


def fake_step(i: int):
    """Deterministic per-batch probabilities (stand-in for the GPU forward; quantised so ties occur)."""
    rs = np.random.RandomState(1000 + i)
    n = 200 if i % 5 else 37            # a short last-batch-like case
    pos = np.round(rs.beta(3, 2, n), 2)
    neg = np.round(rs.beta(2, 3, n), 2)
    return torch.from_numpy(pos), torch.from_numpy(neg)


def cpu_metrics(predicts: torch.Tensor, labels: torch.Tensor):
    from oracle import metrics_oracle as mo          # the checker: stands in for metrics.hip where there is no GPU
    y, s = labels.numpy(), predicts.numpy()
    return mo.average_precision(y, s), mo.roc_auc(y, s)


def test_cpu_metrics_match_sklearn():
    from sklearn.metrics import average_precision_score, roc_auc_score
    for i in range(12):
        pos, neg = fake_step(i)
        y = np.concatenate([np.ones(len(pos)), np.zeros(len(neg))])
        s = np.concatenate([pos.numpy(), neg.numpy()])
        ap, auc = cpu_metrics(torch.from_numpy(s), torch.from_numpy(y))
        assert abs(auc - roc_auc_score(y, s)) < 1e-12 and abs(ap - average_precision_score(y, s)) < 1e-12


def test_shard_steps_give_every_rank_the_same_step_count():
    for n in (0, 1, 7, 60, 237):
        for w in (1, 2, 3, 8):
            plans = [D.shard_steps(n, r, w) for r in range(w)]
            assert len({len(p) for p in plans}) == 1                        # same number of collectives on every rank
            assert sorted(i for p in plans for i in p if i is not None) == list(range(n))


def test_shard_indices_cover_every_batch_once():
    for n in (0, 1, 7, 237):
        for w in (1, 2, 3, 8):
            seen = sorted(i for r in range(w) for i in D.shard_batch_indices(n, r, w))
            assert seen == list(range(n))
            assert all(D.owner_of_batch(i, w) == r for r in range(w) for i in D.shard_batch_indices(n, r, w))
    with pytest.raises(ValueError):
        D.shard_batch_indices(4, 2, 2)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, n_batches: int, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out[rank] = D.evaluate_sharded(fake_step, n_batches, rank, world, metrics_fn=cpu_metrics)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_batches", [9, 1])      # odd count: ranks own different numbers of batches; 1: an idle rank
def test_world_size_2_gloo_matches_single_process(n_batches):
    single = D.evaluate_sharded(fake_step, n_batches, 0, 1, metrics_fn=cpu_metrics)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), n_batches, out), nprocs=2, join=True)
    assert set(out.keys()) == {0, 1}
    for r in (0, 1):
        assert out[r]["num_batches"] == single["num_batches"] == n_batches
        assert abs(out[r]["average_precision"] - single["average_precision"]) < 1e-12
        assert abs(out[r]["roc_auc"] - single["roc_auc"]) < 1e-12


def _grad_worker(rank: int, world: int, port: int, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.Linear(3, 1))
        net[1].bias.requires_grad_(False)                    # a frozen parameter stays out of the bucket
        x = torch.arange(10, dtype=torch.float32).reshape(2, 5) * (rank + 1)
        if rank == 0:
            net(x).sum().backward()
        else:
            net[0](x).sum().backward()                       # rank 1 produces no gradient for net[1].weight
        n = D.allreduce_gradients(net.parameters())
        out[rank] = (n, [None if p.grad is None else p.grad.clone() for p in net.parameters()])
    finally:
        dist.destroy_process_group()


def test_world_size_2_gradient_allreduce_averages_one_flat_bucket():
    """Data-parallel training step: after the all-reduce both ranks hold the mean of the two local gradients."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_grad_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.Linear(3, 1))
    x = torch.arange(10, dtype=torch.float32).reshape(2, 5)
    net(x).sum().backward()
    g0 = [p.grad.clone() for p in net.parameters()]
    net.zero_grad()
    net[0](2 * x).sum().backward()
    g1 = [torch.zeros_like(p) if p.grad is None else p.grad.clone() for p in net.parameters()]
    want = [(a + b) / 2 for a, b in zip(g0, g1)]
    for r in (0, 1):
        n, grads = out[r]
        assert n == 5 * 3 + 3 + 3                            # weights + bias of layer 0, weight of layer 1
        for k in (0, 1, 2):
            assert torch.allclose(grads[k], want[k], atol=1e-6), (r, k)
    assert D.allreduce_gradients(net.parameters()) == 0      # not initialised here: no-op


def _train_worker(rank: int, world: int, port: int, n_batches: int, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = torch.nn.Linear(4, 1)
        opt = torch.optim.SGD(net.parameters(), lr=0.1)

        def step(i):
            x = torch.full((3, 4), float(i + 1))
            return net(x).pow(2).mean()
        losses = D.train_sharded(step, list(net.parameters()), opt, n_batches, rank, world)
        out[rank] = (len(losses), [p.detach().clone() for p in net.parameters()])
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("n_batches", [5, 1])       # odd: rank 1 is one batch short; 1: rank 1 never has a batch
def test_world_size_2_training_loop_with_uneven_batches_does_not_hang(n_batches):
    """with nb % world != 0 the last gradient all-reduce used to be issued by only some ranks (a hang).  Every rank now takes
    ceil(nb / world) steps; after the loop both ranks hold identical parameters."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_train_worker, args=(2, _free_port(), n_batches, out), nprocs=2, join=True)
    assert out[0][0] == (n_batches + 1) // 2 and out[1][0] == n_batches // 2
    for a, b in zip(out[0][1], out[1][1]):
        assert torch.equal(a, b)
