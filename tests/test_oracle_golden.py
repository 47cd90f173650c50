"""Pin the CPU oracle against golden vectors produced by the reference itself
(oracle/make_golden.py).  Integer / index outputs must be bit-exact; fp32 within 1e-4
(BASELINE.json north_star tolerance) — in practice ~1e-6."""
import numpy as np
import pytest
import torch

from oracle import dygformer_oracle as orc
from tests import golden_cases as gc

FP_TOL = 1e-4


@pytest.fixture(scope="module", params=list(gc.CASES))
def case(request):
    c = gc.build_case(request.param)
    g = gc.load_golden(request.param)
    d = c["data"]
    adj = orc.OracleAdjacency(d.src_node_ids, d.dst_node_ids, d.edge_ids, d.node_interact_times)
    return request.param, c, g, adj


def test_recent_sampler_bit_exact(case):
    _, c, g, adj = case
    q_nodes = np.concatenate([c["src"], c["dst"]])
    q_times = np.concatenate([c["times"], c["times"]])
    for k in gc.SAMPLER_KS:
        n, e, t = orc.get_historical_neighbors_recent(adj, q_nodes, q_times, k)
        assert n.dtype == np.int64 and e.dtype == np.int64 and t.dtype == np.float32
        np.testing.assert_array_equal(n, g[f"recent{k}_nbr"])
        np.testing.assert_array_equal(e, g[f"recent{k}_eid"])
        np.testing.assert_array_equal(t, g[f"recent{k}_ts"])


def test_second_hop_float32_query_times(case):
    _, c, g, adj = case
    q_nodes = np.concatenate([c["src"], c["dst"]])
    q_times = np.concatenate([c["times"], c["times"]])
    n, e, t = orc.get_historical_neighbors_recent(adj, q_nodes, q_times, 10)
    n2, e2, t2 = orc.get_historical_neighbors_recent(adj, n.flatten(), t.flatten(), 10)
    np.testing.assert_array_equal(n2, g["hop2_nbr"])
    np.testing.assert_array_equal(e2, g["hop2_eid"])
    np.testing.assert_array_equal(t2, g["hop2_ts"])


def test_windows_and_counts_bit_exact(case):
    _, c, g, adj = case
    cfg = c["cfg"]
    ids = {}
    for tag, q in (("src", c["src"]), ("dst", c["dst"])):
        a, b, cc = orc.get_all_first_hop_neighbors(adj, q, c["times"])
        np.testing.assert_array_equal(np.array([len(x) for x in a]), g[f"{tag}_hist_len"])
        pid, pe, pt = orc.pad_sequences(q, c["times"], a, b, cc, cfg["patch_size"], cfg["max_input_sequence_length"])
        np.testing.assert_array_equal(pid, g[f"{tag}_pad_ids"])
        np.testing.assert_array_equal(pe, g[f"{tag}_pad_eids"])
        np.testing.assert_array_equal(pt, g[f"{tag}_pad_times"])
        ids[tag] = pid
    cs, cd = orc.count_nodes_appearances(ids["src"], ids["dst"])
    np.testing.assert_array_equal(cs, g["src_counts"])
    np.testing.assert_array_equal(cd, g["dst_counts"])


def test_forward_stages_and_embeddings(case):
    name, c, g, adj = case
    cfg = c["cfg"]
    taps = {}
    with torch.no_grad():
        se, de = orc.dygformer_forward(c["params"], c["node_feat"], c["edge_feat"], adj, c["src"], c["dst"], c["times"],
                                       cfg["patch_size"], cfg["max_input_sequence_length"], cfg["num_heads"],
                                       cfg["num_layers"], taps=taps)
    R = gc.TAP_ROWS
    np.testing.assert_allclose(taps["encoder_input"][:R].numpy(), g["encoder_input_rows"], atol=FP_TOL, rtol=0)
    for l, x in enumerate(taps["layer_outputs"]):
        np.testing.assert_allclose(x[:R].numpy(), g[f"layer{l}_rows"], atol=FP_TOL, rtol=0)
    np.testing.assert_allclose(se.numpy(), g["src_emb"], atol=FP_TOL, rtol=0)
    np.testing.assert_allclose(de.numpy(), g["dst_emb"], atol=FP_TOL, rtol=0)


def test_link_prediction_step(case):
    _, c, g, adj = case
    cfg = c["cfg"]
    pos, neg = orc.link_prediction_step(c["params"], c["mparams"], c["node_feat"], c["edge_feat"], adj, c["src"], c["dst"],
                                        c["neg_dst"], c["times"], cfg["patch_size"], cfg["max_input_sequence_length"])
    np.testing.assert_allclose(pos.numpy(), g["pos_prob"], atol=FP_TOL, rtol=0)
    np.testing.assert_allclose(neg.numpy(), g["neg_prob"], atol=FP_TOL, rtol=0)


def test_golden_covers_edge_cases():
    """The fixtures must actually contain the edge cases SURVEY.md §8(c) lists."""
    g = gc.load_golden("bip_p2_l64")
    assert (g["src_hist_len"] == 0).any() or (g["dst_hist_len"] == 0).any()       # empty history
    assert (g["src_hist_len"] > 63).any() or (g["dst_hist_len"] > 63).any()       # history > L-1
    assert (g["src_counts"] > 1).any()                                            # co-occurrence > 1
    assert (g["src_pad_ids"] == 0).any()                                          # padding present
    h = gc.load_golden("hub_p4_l48")
    assert h["src_pad_ids"].shape[1] != h["dst_pad_ids"].shape[1]                 # S_src != S_dst
    n = gc.load_golden("gen_p1_l32")
    assert len(n["src_emb"]) % 2 == 1                                             # odd / short batch
