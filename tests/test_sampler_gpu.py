"""GPU parity (through the C ABI) of the temporal neighbour lookup: bit-exact against the golden
vectors of the reference and against the oracle on larger seeded graphs."""
import numpy as np
import pytest
import torch

from dyglib_amd import synthetic as syn
from oracle import dygformer_oracle as orc
from tests import golden_cases as gc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=list(gc.CASES))
def case(request):
    from dyglib_amd import get_neighbor_sampler
    c = gc.build_case(request.param)
    g = gc.load_golden(request.param)
    sampler = get_neighbor_sampler(c["data"], "recent", seed=1, device="cuda:0")
    return c, g, sampler


def test_recent_matches_golden(case):
    c, g, s = case
    q_nodes = np.concatenate([c["src"], c["dst"]])
    q_times = np.concatenate([c["times"], c["times"]])
    for k in gc.SAMPLER_KS:
        n, e, t = s.get_historical_neighbors(q_nodes, q_times, num_neighbors=k)
        assert n.dtype == np.int64 and e.dtype == np.int64 and t.dtype == np.float32 and n.shape == (len(q_nodes), k)
        np.testing.assert_array_equal(n, g[f"recent{k}_nbr"])
        np.testing.assert_array_equal(e, g[f"recent{k}_eid"])
        np.testing.assert_array_equal(t, g[f"recent{k}_ts"])


def test_multi_hop_float32_times_match_golden(case):
    c, g, s = case
    q_nodes = np.concatenate([c["src"], c["dst"]])
    q_times = np.concatenate([c["times"], c["times"]])
    ids, eids, ts = s.get_multi_hop_neighbors(2, q_nodes, q_times, num_neighbors=10)
    np.testing.assert_array_equal(ids[0], g["recent10_nbr"])
    np.testing.assert_array_equal(ids[1].reshape(-1, 10), g["hop2_nbr"])
    np.testing.assert_array_equal(eids[1].reshape(-1, 10), g["hop2_eid"])
    np.testing.assert_array_equal(ts[1].reshape(-1, 10), g["hop2_ts"])


def test_first_hop_and_windows_match_golden(case):
    c, g, s = case
    cfg = c["cfg"]
    for tag, q in (("src", c["src"]), ("dst", c["dst"])):
        a, b, cc = s.get_all_first_hop_neighbors(q, c["times"])
        np.testing.assert_array_equal(np.array([len(x) for x in a]), g[f"{tag}_hist_len"])
        assert a[0].dtype == np.int64 and b[0].dtype == np.int64 and cc[0].dtype == np.float64
        pid, pe, pt = s.padded_windows(q, c["times"], cfg["patch_size"], cfg["max_input_sequence_length"])
        np.testing.assert_array_equal(pid, g[f"{tag}_pad_ids"])
        np.testing.assert_array_equal(pe, g[f"{tag}_pad_eids"])
        np.testing.assert_array_equal(pt, g[f"{tag}_pad_times"])
        # the lists themselves: same content as the oracle's
        adj = orc.OracleAdjacency(c["data"].src_node_ids, c["data"].dst_node_ids, c["data"].edge_ids, c["data"].node_interact_times)
        oa, ob, oc = orc.get_all_first_hop_neighbors(adj, q, c["times"])
        for x, y in zip(a + b + cc, oa + ob + oc):
            np.testing.assert_array_equal(x, y)


def test_cooccurrence_matches_golden(case):
    from dyglib_amd import count_nodes_appearances
    c, g, s = case
    cs, cd = count_nodes_appearances(g["src_pad_ids"], g["dst_pad_ids"], device="cuda:0")
    np.testing.assert_array_equal(cs.cpu().numpy(), g["src_counts"])
    np.testing.assert_array_equal(cd.cpu().numpy(), g["dst_counts"])


def test_large_graph_against_oracle_and_edge_cases():
    """Wikipedia-scale rows (degree up to thousands: exercises the multi-level 64-ary search),
    queries at times equal to stored timestamps, before the first and after the last interaction,
    node 0, k larger than any history."""
    from dyglib_amd import get_neighbor_sampler
    data, _, _ = syn.make_bipartite_graph(300, 40, 60000, seed=3, duplicate_time_every=5, edge_feat_dim=4)
    s = get_neighbor_sampler(data, "recent", seed=0, device="cuda:0")
    adj = orc.OracleAdjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    deg = np.diff(adj.indptr)
    assert deg.max() > 64 * 64                      # at least 3 search levels somewhere
    rs = np.random.RandomState(0)
    idx = rs.randint(0, data.num_interactions, 3000)
    nodes = np.concatenate([data.src_node_ids[idx], data.dst_node_ids[idx], [0, 0, 1, int(np.argmax(deg))]])
    times = np.concatenate([data.node_interact_times[idx], data.node_interact_times[idx], [0.0, 1e9, -5.0, 1e9]])
    for k in (1, 7, 64, 100):
        n, e, t = s.get_historical_neighbors(nodes, times, num_neighbors=k)
        on, oe, ot = orc.get_historical_neighbors_recent(adj, nodes, times, k)
        np.testing.assert_array_equal(n, on)
        np.testing.assert_array_equal(e, oe)
        np.testing.assert_array_equal(t, ot)
    for L, P in ((64, 2), (33, 4), (512, 8)):
        pid, pe, pt = s.padded_windows(nodes, times, P, L)
        oid, oe, ot = orc.first_hop_windows(adj, nodes, times, P, L)
        np.testing.assert_array_equal(pid, oid)
        np.testing.assert_array_equal(pe, oe)
        np.testing.assert_array_equal(pt, ot)
    # properties at full size: outputs sorted by time inside a row, all strictly earlier than the query
    n, e, t = s.get_historical_neighbors(nodes, times, num_neighbors=20)
    valid = n != 0
    assert (np.diff(t, axis=1)[valid[:, 1:] & valid[:, :-1]] >= 0).all()
    assert (t[valid].astype(np.float64) <= times[:, None].repeat(20, 1)[valid] + 0.25).all()
    # empty batch and invalid k
    z = s.get_historical_neighbors(np.zeros(0, dtype=np.int64), np.zeros(0), num_neighbors=5)
    assert z[0].shape == (0, 5)
    with pytest.raises(AssertionError):
        s.get_historical_neighbors(nodes, times, num_neighbors=0)
    with pytest.raises(AssertionError):
        s.padded_windows(nodes, times, 1, 1)


def test_four_queries_per_wave_form_matches_oracle():
    """Batches of 16,384 queries and more take the 16-lanes-per-query kernels (four searches per wave, sampler.hip): same rows,
    bit for bit — on a graph with rows from empty to tens of thousands of entries, duplicate timestamps, queries exactly at stored
    timestamps, and a batch size that is not a multiple of four (idle sub-groups in the last wave)."""
    from dyglib_amd import get_neighbor_sampler
    data, _, _ = syn.make_bipartite_graph(300, 40, 60000, seed=3, duplicate_time_every=5, edge_feat_dim=4)
    s = get_neighbor_sampler(data, "recent", seed=0, device="cuda:0")
    adj = orc.OracleAdjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    rs = np.random.RandomState(4)
    idx = rs.randint(0, data.num_interactions, 10001)
    nodes = np.concatenate([data.src_node_ids[idx], data.dst_node_ids[idx], [0, 0, 1]])          # 20,005 queries
    times = np.concatenate([data.node_interact_times[idx], data.node_interact_times[idx] + rs.choice([0.0, 1e-3, -1e-3], len(idx)), [0.0, 1e9, -5.0]])
    assert len(nodes) >= 16384 and len(nodes) % 4 != 0
    for k in (1, 20, 33):
        n, e, t = s.get_historical_neighbors(nodes, times, num_neighbors=k)
        on, oe, ot = orc.get_historical_neighbors_recent(adj, nodes, times, k)
        np.testing.assert_array_equal(n, on)
        np.testing.assert_array_equal(e, oe)
        np.testing.assert_array_equal(t, ot)
    pid, pe, pt = s.padded_windows(nodes, times, 2, 64)
    oid, oe, ot = orc.first_hop_windows(adj, nodes, times, 2, 64)
    np.testing.assert_array_equal(pid, oid)
    np.testing.assert_array_equal(pe, oe)
    np.testing.assert_array_equal(pt, ot)


def test_cooccurrence_random_rows_against_oracle():
    from dyglib_amd import count_nodes_appearances
    rs = np.random.RandomState(1)
    for (B, Ss, Sd, hi) in ((64, 64, 64, 6), (17, 5, 64, 30), (3, 512, 512, 40), (9, 1, 1, 2)):
        s = rs.randint(0, hi, size=(B, Ss)).astype(np.int64)
        d = rs.randint(0, hi, size=(B, Sd)).astype(np.int64)
        cs, cd = count_nodes_appearances(s, d, device="cuda:0")
        os_, od = orc.count_nodes_appearances(s, d)
        np.testing.assert_array_equal(cs.cpu().numpy(), os_)
        np.testing.assert_array_equal(cd.cpu().numpy(), od)
