"""Evaluation metrics (reference utils/metrics.py:5-34 -> scikit-learn; BCELoss of evaluate_models_utils.py:145).
CPU: the oracle restatement against the reference's own outputs (tests/golden/metrics.npz).  GPU: dygnn_link_metrics
against those fixtures and against the oracle on seeded inputs.  Tolerances: AP / AUC 1e-12 absolute (float64; the rank
counts are exact, only the summation order differs from numpy's pairwise sum); BCELoss 1e-6 relative (torch reduces
float32 terms in float32, the kernel in float64)."""
import numpy as np
import pytest

from tests import golden_cases as gc
from oracle import metrics_oracle as mo

TOL = 1e-12
CASES = list(gc.METRIC_CASES)


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_fixture(name):
    g = gc.load_golden("metrics")
    p, y = gc.build_metric_case(name)
    assert abs(mo.average_precision(y, p) - float(g[f"{name}|average_precision"])) <= TOL
    assert abs(mo.roc_auc(y, p) - float(g[f"{name}|roc_auc"])) <= TOL
    assert abs(mo.roc_auc(y, p) - float(g[f"{name}|node_roc_auc"])) <= TOL
    assert abs(mo.bce_loss(y, p) - float(g[f"{name}|bce_loss"])) <= 1e-6 * max(1.0, float(g[f"{name}|bce_loss"]))


def test_oracle_single_class_raises():
    with pytest.raises(ValueError):
        mo.roc_auc(np.ones(5, dtype=np.float32), np.linspace(0, 1, 5).astype(np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_device_metrics_match_reference_fixture(name):
    import torch
    from dyglib_amd import get_link_prediction_metrics, get_node_classification_metrics, link_prediction_metrics_device
    g = gc.load_golden("metrics")
    p, y = gc.build_metric_case(name)
    pt, yt = torch.from_numpy(p).cuda(), torch.from_numpy(y).cuda()
    m = get_link_prediction_metrics(predicts=pt, labels=yt)
    assert set(m) == {"average_precision", "roc_auc"} and all(isinstance(v, float) for v in m.values())
    assert abs(m["average_precision"] - float(g[f"{name}|average_precision"])) <= TOL
    assert abs(m["roc_auc"] - float(g[f"{name}|roc_auc"])) <= TOL
    assert abs(get_node_classification_metrics(predicts=pt, labels=yt)["roc_auc"] - float(g[f"{name}|node_roc_auc"])) <= TOL
    _, _, loss, _ = link_prediction_metrics_device(pt, yt)
    ref = float(g[f"{name}|bce_loss"])
    assert abs(float(loss.item()) - ref) <= 1e-6 * max(1.0, ref)


@pytest.mark.gpu
def test_device_metrics_batched_groups_equal_single_calls_and_oracle():
    """32 evaluation batches in one launch: every group is bit-identical to its own call and within 1e-12 of the oracle;
    the order of the samples inside a group does not matter (rank statistics)."""
    import torch
    from dyglib_amd import link_prediction_metrics_device
    rs = np.random.RandomState(3)
    G, n = 32, 400
    y = np.tile(np.concatenate([np.ones(200), np.zeros(200)]).astype(np.float32), (G, 1))
    p = (1 / (1 + np.exp(-(rs.standard_normal((G, n)) + y)))).astype(np.float32)
    p[5] = np.round(p[5] * 10) / 10                                  # ties
    ap, auc, loss, status = (t.cpu().numpy() for t in link_prediction_metrics_device(torch.from_numpy(p).cuda(), torch.from_numpy(y).cuda()))
    assert (status == 0).all()
    for g in range(G):
        assert abs(ap[g] - mo.average_precision(y[g], p[g])) <= TOL
        assert abs(auc[g] - mo.roc_auc(y[g], p[g])) <= TOL
        assert abs(loss[g] - mo.bce_loss(y[g], p[g])) <= 1e-6
    one = [t.cpu().numpy() for t in link_prediction_metrics_device(torch.from_numpy(p[7]).cuda(), torch.from_numpy(y[7]).cuda())]
    assert one[0][0] == ap[7] and one[1][0] == auc[7] and one[2][0] == loss[7]
    perm = rs.permutation(n)
    sh = [t.cpu().numpy() for t in link_prediction_metrics_device(torch.from_numpy(p[7][perm].copy()).cuda(), torch.from_numpy(y[7][perm].copy()).cuda())]
    assert abs(sh[0][0] - ap[7]) <= TOL and sh[1][0] == auc[7]


@pytest.mark.gpu
def test_device_metrics_large_split_and_ragged_sizes():
    """a whole evaluation split in one call (node classification: evaluate_models_utils.py:245-249), and sizes around the
    256-sample block / 1024-sample tile boundaries"""
    import torch
    from dyglib_amd import get_link_prediction_metrics
    rs = np.random.RandomState(4)
    for n in (1, 2, 255, 256, 257, 1023, 1024, 1025, 60000):
        y = (rs.random_sample(n) < 0.3).astype(np.float32)
        y[0], y[-1] = 1.0, 0.0
        if n == 1:
            with pytest.raises(ValueError):
                get_link_prediction_metrics(torch.rand(1).cuda(), torch.ones(1).cuda())
            continue
        p = np.round(rs.random_sample(n) * 997).astype(np.float32) / np.float32(997) if n > 2000 else rs.random_sample(n).astype(np.float32)
        m = get_link_prediction_metrics(torch.from_numpy(p).cuda(), torch.from_numpy(y).cuda())
        assert abs(m["average_precision"] - mo.average_precision(y, p)) <= TOL, n
        assert abs(m["roc_auc"] - mo.roc_auc(y, p)) <= TOL, n


@pytest.mark.gpu
def test_device_metrics_single_class_raises_like_sklearn():
    import torch
    from dyglib_amd import get_link_prediction_metrics, link_prediction_metrics_device
    p = torch.rand(10).cuda()
    for y in (torch.ones(10), torch.zeros(10)):
        with pytest.raises(ValueError, match="Only one class present"):
            get_link_prediction_metrics(p, y.cuda())
    ap, auc, _, status = link_prediction_metrics_device(p, torch.ones(10).cuda())
    assert int(status.item()) == 1 and bool(torch.isnan(auc).item()) and float(ap.item()) == 1.0


def test_metrics_refuse_cpu_tensors():
    import torch
    from dyglib_amd import _capi, get_link_prediction_metrics
    with pytest.raises(_capi.DygnnError):
        get_link_prediction_metrics(torch.rand(4), torch.tensor([1.0, 0.0, 1.0, 0.0]))
