"""Property tests (hypothesis) of the host side of the path: the C++ CSR builder (dygnn_csr_build_host) against the oracle's
restatement of the reference's list-of-lists build (utils/utils.py:283-302, :96-103), and the oracle's sampler against a
brute-force statement of the history rule — on adversarial small graphs: duplicate timestamps, non-chronological edge
order, self interactions, isolated nodes (empty rows), edges that touch the padding node 0.  Also the host-side id checks
(IndexError like the reference's list / tensor indexing).  No kernel is launched here."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from dyglib_amd.temporal_csr import TemporalCSR
from oracle import dygformer_oracle as orc


@st.composite
def graphs(draw):
    n_nodes = draw(st.integers(1, 12))
    n_edges = draw(st.integers(0, 40))
    lo = draw(st.sampled_from([0, 1]))                      # 0: some edges may touch the padding node
    ids = st.integers(lo, n_nodes)
    src = np.array(draw(st.lists(ids, min_size=n_edges, max_size=n_edges)), dtype=np.int64)
    dst = np.array(draw(st.lists(ids, min_size=n_edges, max_size=n_edges)), dtype=np.int64)
    # few distinct times -> many ties; drawn unsorted -> the per-row stable sort matters
    ts = np.array(draw(st.lists(st.sampled_from([0.0, 1.0, 1.0, 2.5, 2.5, 2.5, 7.0, 1e6, 2.68e6]), min_size=n_edges, max_size=n_edges)), dtype=np.float64)
    eid = np.arange(1, n_edges + 1, dtype=np.int64)
    return n_nodes, src, dst, eid, ts


@settings(max_examples=150, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(graphs())
def test_csr_builder_equals_reference_build(g):
    n_nodes, src, dst, eid, ts = g
    csr = TemporalCSR.from_interactions(src, dst, eid, ts, num_nodes=n_nodes + 1)
    # the reference build, literally: append under both endpoints in edge-list order, stable sort by time (utils/utils.py:293-300, :98-100)
    adj = [[] for _ in range(n_nodes + 1)]
    for s, d, e, t in zip(src.tolist(), dst.tolist(), eid.tolist(), ts.tolist()):
        adj[s].append((d, e, t))
        adj[d].append((s, e, t))
    for v in range(n_nodes + 1):
        row = sorted(adj[v], key=lambda x: x[2])
        a, b = int(csr.indptr[v]), int(csr.indptr[v + 1])
        assert b - a == len(row)
        assert csr.nbr[a:b].tolist() == [x[0] for x in row]
        assert csr.eid[a:b].tolist() == [x[1] for x in row]
        assert csr.ts[a:b].tolist() == [x[2] for x in row]
    assert int(csr.indptr[-1]) == 2 * len(src)
    if len(src):                                            # the oracle's own adjacency is the same structure
        oa = orc.OracleAdjacency(src, dst, eid, ts)
        m = min(len(oa.indptr), len(csr.indptr))
        np.testing.assert_array_equal(oa.indptr[:m], csr.indptr[:m])
        np.testing.assert_array_equal(oa.nbr, csr.nbr)
        np.testing.assert_array_equal(oa.eid, csr.eid)


@settings(max_examples=150, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(graphs(), st.integers(1, 6), st.data())
def test_oracle_recent_sampler_is_the_strictly_earlier_last_k(g, k, data):
    n_nodes, src, dst, eid, ts = g
    if len(src) == 0:
        return
    adj = orc.OracleAdjacency(src, dst, eid, ts)
    n_rows = len(adj.indptr) - 1
    q_nodes = np.array(data.draw(st.lists(st.integers(0, n_rows - 1), min_size=1, max_size=8)), dtype=np.int64)
    q_times = np.array(data.draw(st.lists(st.sampled_from([0.0, 1.0, 2.5, 2.5000001, 7.0, 1e6, 3e6]), min_size=len(q_nodes), max_size=len(q_nodes))))
    nbr, ed, tm = orc.get_historical_neighbors_recent(adj, q_nodes, q_times, k)
    assert nbr.shape == (len(q_nodes), k) and nbr.dtype == np.int64 and ed.dtype == np.int64 and tm.dtype == np.float32     # utils/utils.py:161-167
    for r, (v, t) in enumerate(zip(q_nodes.tolist(), q_times.tolist())):
        a, b = int(adj.indptr[v]), int(adj.indptr[v + 1])
        hist = [(int(adj.nbr[p]), int(adj.eid[p]), float(adj.ts[p])) for p in range(a, b) if adj.ts[p] < t]      # strictly earlier (:139-141)
        last = hist[-k:]
        pad = k - len(last)
        assert nbr[r, :pad].tolist() == [0] * pad and ed[r, :pad].tolist() == [0] * pad and tm[r, :pad].tolist() == [0.0] * pad    # right-aligned (:207-209)
        assert nbr[r, pad:].tolist() == [x[0] for x in last]
        assert ed[r, pad:].tolist() == [x[1] for x in last]
        assert tm[r, pad:].tolist() == [float(np.float32(x[2])) for x in last]
        assert all(tm[r, pad:][i] <= tm[r, pad:][i + 1] for i in range(len(last) - 1))                               # ascending in time


def test_host_id_checks_raise_index_error_like_the_reference():
    csr = TemporalCSR.from_interactions(np.array([1, 2]), np.array([3, 3]), np.array([1, 2]), np.array([0.0, 1.0]))
    assert csr.num_nodes == 4 and csr.max_ids() == (3, 2)
    csr.check_query_ids(np.array([0, 3]))
    csr.check_query_ids([])                                  # empty batch
    for bad in ([4], [-1], [1, 2, 99]):
        with pytest.raises(IndexError):
            csr.check_query_ids(np.array(bad))
    with pytest.raises(IndexError):
        csr.check_query_ids(np.array([3]), limit=3)          # a feature table shorter than the graph
    csr.check_tables(num_node_rows=4, num_edge_rows=3)
    with pytest.raises(IndexError):
        csr.check_tables(num_node_rows=3, num_edge_rows=3)   # neighbour id 3 has no feature row
    with pytest.raises(IndexError):
        csr.check_tables(num_node_rows=4, num_edge_rows=2)   # edge id 2 has no feature row
    # the numpy-input sampler API raises before anything reaches a kernel (utils/utils.py:139: list index out of range)
    from dyglib_amd import NeighborSampler
    s = NeighborSampler(None, "recent", seed=0, csr=csr, device="cpu")
    with pytest.raises(IndexError):
        s.get_historical_neighbors(np.array([7]), np.array([1.0]), 3)
    with pytest.raises(IndexError):
        s.get_all_first_hop_neighbors(np.array([-2]), np.array([1.0]))
