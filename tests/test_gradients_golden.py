"""Parameter gradients against fixtures produced by the REFERENCE model's own autograd (oracle/make_golden.py ->
tests/golden/grads_*.npz; eval mode so dropout is the identity).  CPU: the oracle's autograd is pinned to them.
GPU: the hand-written HIP backward pass is checked against them directly.  Tolerance 1e-4 * max(1, max|gradient|) per
tensor for stored elements; the random projections of the big matrices (sums over up to 160k products) get the same
relative bound on the projection value."""
import numpy as np
import pytest
import torch

from oracle import dygformer_oracle as orc
from tests import golden_cases as gc
from tests.parity import close_scaled

TOL = 1e-4


def _check(name, grads: dict, g: dict):
    for k, got in grads.items():
        got = np.asarray(got)
        if f"{k}|full" in g:
            ref = g[f"{k}|full"]
            close_scaled(got, ref, f"{name} grad {k}", label=f"gradients vs reference autograd, {name} (worst full tensor, scaled bar)")
        else:
            absmax = float(g[f"{k}|absmax"])
            tol = TOL * max(1.0, absmax)
            assert np.abs(got[:8, :8] - g[f"{k}|corner"]).max() <= tol, (name, k)
            for i, want in enumerate(g[f"{k}|proj"]):
                r = np.random.RandomState(gc.GRAD_SEED + 1 + i).standard_normal(got.shape)
                val = float((got.astype(np.float64) * r).sum())
                # a projection sums `size` products: bound its error by tol * sqrt(size) (random signs)
                assert abs(val - want) <= tol * np.sqrt(got.size), (name, k, i, val, float(want))


@pytest.mark.parametrize("name", gc.GRAD_CASES)
def test_oracle_autograd_matches_reference_gradients(name):
    c = gc.build_case(name)
    g = gc.load_golden("grads_" + name)
    cfg = c["cfg"]
    params = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in c["params"].items()}
    d = c["data"]
    adj = orc.OracleAdjacency(d.src_node_ids, d.dst_node_ids, d.edge_ids, d.node_interact_times)
    s, t = orc.dygformer_forward(params, c["node_feat"], c["edge_feat"], adj, c["src"], c["dst"], c["times"], cfg["patch_size"],
                                 cfg["max_input_sequence_length"], cfg["num_heads"], cfg["num_layers"])
    G1, G2 = gc.grad_loss_weights(len(c["src"]))
    loss = (s * torch.from_numpy(G1)).sum() + (t * torch.from_numpy(G2)).sum()
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-3 * max(1.0, abs(float(g["loss"])))
    _check(name, {k: v.grad.numpy() for k, v in params.items()}, g)


@pytest.mark.gpu
@pytest.mark.parametrize("name", gc.GRAD_CASES)
def test_hip_backward_matches_reference_gradients(name):
    from tests.test_dygformer_gpu import build_model
    c = gc.build_case(name)
    g = gc.load_golden("grads_" + name)
    model, _ = build_model(c)
    model.eval()
    s, t = model.compute_src_dst_node_temporal_embeddings(c["src"], c["dst"], c["times"])
    G1, G2 = gc.grad_loss_weights(len(c["src"]))
    loss = (s * torch.from_numpy(G1).cuda()).sum() + (t * torch.from_numpy(G2).cuda()).sum()
    loss.backward()
    _check(name, {k: p.grad.detach().cpu().numpy() for k, p in model.named_parameters()}, g)
