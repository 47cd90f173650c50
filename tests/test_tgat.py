"""TGAT (BASELINE config 3): the oracle restatement against reference-generated golden vectors (CPU), and the HIP
path through the C ABI against golden + oracle (GPU).  fp32 tolerance: plain absolute 1e-4 (tests/parity.py)."""
import numpy as np
import pytest
import torch

from dyglib_amd import synthetic as syn
from oracle import dygformer_oracle as orc
from oracle import tgat_oracle as torc
from tests import golden_cases as gc
from tests.parity import close  # plain 1e-4 absolute; observed errors are printed at the end of the run

TOL = 1e-4




@pytest.fixture(scope="module", params=list(gc.TGAT_CASES))
def case(request):
    c = gc.build_tgat_case(request.param)
    g = gc.load_golden(request.param)
    d = c["data"]
    adj = orc.OracleAdjacency(d.src_node_ids, d.dst_node_ids, d.edge_ids, d.node_interact_times)
    return request.param, c, g, adj


def test_oracle_matches_reference_golden(case):
    name, c, g, adj = case
    cfg = c["tgat_cfg"]
    s, d = torc.tgat_forward(c["tgat_params"], c["node_feat"], c["edge_feat"], adj, c["src"], c["dst"], c["times"],
                             cfg["num_layers"], cfg["num_neighbors"], cfg["num_heads"])
    close(s.numpy(), g["src_emb"], name + " src")
    close(d.numpy(), g["dst_emb"], name + " dst")
    ns, nd = torc.tgat_forward(c["tgat_params"], c["node_feat"], c["edge_feat"], adj, c["src"], c["neg_dst"], c["times"],
                               cfg["num_layers"], cfg["num_neighbors"], cfg["num_heads"])
    close(nd.numpy(), g["neg_dst_emb"], name + " neg dst")


def test_state_dict_keys_match_reference():
    from dyglib_amd import TGAT, NeighborSampler
    from dyglib_amd.temporal_csr import TemporalCSR
    data, nf, ef = syn.make_bipartite_graph(5, 3, 20, seed=0)
    sampler = NeighborSampler(None, "recent", seed=0, csr=TemporalCSR.from_interactions(
        data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times), device="cpu")
    m = TGAT(nf, ef, sampler, time_feat_dim=100, num_layers=2, num_heads=2, dropout=0.1, device="cpu")
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == syn.tgat_param_shapes(num_layers=2)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_tgat_params(1).items()}, strict=True)


def _model(c, device="cuda:0"):
    from dyglib_amd import TGAT, get_neighbor_sampler
    cfg = c["tgat_cfg"]
    sampler = get_neighbor_sampler(c["data"], "recent", seed=1, device=device)
    m = TGAT(c["node_feat"], c["edge_feat"], sampler, time_feat_dim=cfg["time_feat_dim"], num_layers=cfg["num_layers"],
             num_heads=cfg["num_heads"], dropout=0.1, device=device)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in c["tgat_params"].items()}, strict=True)
    return m.to(device).eval()


@pytest.mark.gpu
def test_hip_matches_golden(case):
    name, c, g, adj = case
    m = _model(c)
    k = c["tgat_cfg"]["num_neighbors"]
    with torch.no_grad():
        s, d = m.compute_src_dst_node_temporal_embeddings(c["src"], c["dst"], c["times"], num_neighbors=k)
        ns, nd = m.compute_src_dst_node_temporal_embeddings(c["src"], c["neg_dst"], c["times"], num_neighbors=k)
        s2, d2 = m.compute_src_dst_node_temporal_embeddings(c["src"], c["dst"], c["times"], num_neighbors=k)
    assert s.shape == (len(c["src"]), 172) and s.dtype == torch.float32 and s.is_cuda
    close(s.cpu().numpy(), g["src_emb"], name + " src")
    close(d.cpu().numpy(), g["dst_emb"], name + " dst")
    close(ns.cpu().numpy(), g["neg_src_emb"], name + " neg src")
    close(nd.cpu().numpy(), g["neg_dst_emb"], name + " neg dst")
    assert torch.equal(s, s2) and torch.equal(d, d2)
    with pytest.raises(AssertionError), torch.no_grad():
        m.compute_src_dst_node_temporal_embeddings(c["src"], c["dst"], c["times"], num_neighbors=0)     # utils/utils.py:157
    with pytest.raises(NotImplementedError):       # autograd recording: no silent graph-less result (the HIP path has no TGAT backward)
        m.compute_src_dst_node_temporal_embeddings(c["src"], c["dst"], c["times"], num_neighbors=k)
    with pytest.raises(IndexError), torch.no_grad():   # out-of-range id on the numpy API: IndexError like the reference's indexing
        m.compute_src_dst_node_temporal_embeddings(np.array([10 ** 6]), c["dst"][:1], c["times"][:1], num_neighbors=k)
    # models/TGAT.py:66-136: the layer-wise entry point; layer 0 = raw features, top layer = the source side of the pair call
    with torch.no_grad():
        e0 = m.compute_node_temporal_embeddings(c["src"], c["times"], current_layer_num=0, num_neighbors=k)
        eL = m.compute_node_temporal_embeddings(c["src"], c["times"], current_layer_num=m.num_layers, num_neighbors=k)
    assert torch.equal(e0.cpu(), torch.from_numpy(c["node_feat"])[torch.from_numpy(c["src"])])
    assert torch.equal(eL, s)


@pytest.mark.gpu
def test_hip_reddit_shape_batch_against_oracle():
    """BASELINE config 3 shape: k=20, 2 layers, B=200 on a mid-size graph, vs the oracle."""
    from dyglib_amd import TGAT, get_neighbor_sampler
    data, nf, ef = syn.make_bipartite_graph(500, 60, 30000, seed=31)
    nf[1:] = np.random.RandomState(5).standard_normal(nf[1:].shape).astype(np.float32) * 0.5
    params = syn.make_tgat_params(9)
    sampler = get_neighbor_sampler(data, "recent", seed=1, device="cuda:0")
    m = TGAT(nf, ef, sampler, 100, num_layers=2, num_heads=2, dropout=0.1, device="cuda:0")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    m = m.to("cuda:0").eval()
    adj = orc.OracleAdjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    idx = np.arange(data.num_interactions - 200, data.num_interactions)
    src, dst, t = data.src_node_ids[idx], data.dst_node_ids[idx], data.node_interact_times[idx]
    os_, od = torc.tgat_forward(params, nf, ef, adj, src, dst, t, 2, 20, 2)
    with torch.no_grad():
        gs, gd = m.compute_src_dst_node_temporal_embeddings(src, dst, t, num_neighbors=20)
    close(gs.cpu().numpy(), os_.numpy(), "src")
    close(gd.cpu().numpy(), od.numpy(), "dst")


@pytest.mark.gpu
def test_hip_rows_do_not_depend_on_the_batch_they_are_in():
    """TGAT has no batch-dependent padding (fixed k): eight 200-edge evaluation steps in one call give the rows of the
    eight separate calls (what tools/bench_tgat.py --fuse-steps and evaluate.py rely on)."""
    from dyglib_amd import TGAT, get_neighbor_sampler
    data, nf, ef = syn.make_bipartite_graph(500, 60, 30000, seed=31)
    sampler = get_neighbor_sampler(data, "recent", seed=1, device="cuda:0")
    m = TGAT(nf, ef, sampler, 100, num_layers=2, num_heads=2, dropout=0.1, device="cuda:0")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_tgat_params(9).items()})
    m = m.to("cuda:0").eval()
    idx = np.arange(data.num_interactions - 1600, data.num_interactions)
    src, dst, t = data.src_node_ids[idx], data.dst_node_ids[idx], data.node_interact_times[idx]
    with torch.no_grad():
        fs, fd = m.compute_src_dst_node_temporal_embeddings(src, dst, t, num_neighbors=20)
        for j in range(8):
            sl = slice(200 * j, 200 * (j + 1))
            ss, sd = m.compute_src_dst_node_temporal_embeddings(src[sl], dst[sl], t[sl], num_neighbors=20)
            assert torch.equal(fs[sl], ss) and torch.equal(fd[sl], sd), j


@pytest.mark.gpu
def test_hip_rows_do_not_depend_on_the_rows_per_workgroup(monkeypatch):
    """the row-block chains (tgat_chain.hip: the layer form of TGN calls, DYGNN_TGAT_CHAIN=1 selects it for TGAT) give a workgroup 4, 8, 16 or
    32 rows depending on the level size; a row's bits are the same for every choice (DYGNN_CHAIN_MT forces one), and TGAT's own
    product-by-product GEMM form agrees to rounding"""
    from dyglib_amd import TGAT, get_neighbor_sampler
    data, nf, ef = syn.make_bipartite_graph(500, 60, 30000, seed=31)
    nf[1:] = np.random.RandomState(5).standard_normal(nf[1:].shape).astype(np.float32) * 0.5
    sampler = get_neighbor_sampler(data, "recent", seed=1, device="cuda:0")
    m = TGAT(nf, ef, sampler, 100, num_layers=2, num_heads=2, dropout=0.1, device="cuda:0")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_tgat_params(9).items()})
    m = m.to("cuda:0").eval()
    idx = np.arange(data.num_interactions - 203, data.num_interactions)          # 406 roots: not a multiple of any block size
    src, dst, t = data.src_node_ids[idx], data.dst_node_ids[idx], data.node_interact_times[idx]
    outs = {}
    with torch.no_grad():
        monkeypatch.setenv("DYGNN_TGAT_CHAIN", "1")
        for mt in ("1", "2", "4", "8"):
            monkeypatch.setenv("DYGNN_CHAIN_MT", mt)
            outs[mt] = m.compute_src_dst_node_temporal_embeddings(src, dst, t, num_neighbors=20)
        monkeypatch.delenv("DYGNN_CHAIN_MT")
        auto = m.compute_src_dst_node_temporal_embeddings(src, dst, t, num_neighbors=20)
        monkeypatch.delenv("DYGNN_TGAT_CHAIN")
        gemm = m.compute_src_dst_node_temporal_embeddings(src, dst, t, num_neighbors=20)
    for mt in ("2", "4", "8"):
        assert torch.equal(outs["1"][0], outs[mt][0]) and torch.equal(outs["1"][1], outs[mt][1]), mt
    assert torch.equal(outs["1"][0], auto[0]) and torch.equal(outs["1"][1], auto[1])
    close(auto[0].cpu().numpy(), gemm[0].cpu().numpy(), "chain vs GEMM path src")
    close(auto[1].cpu().numpy(), gemm[1].cpu().numpy(), "chain vs GEMM path dst")


@pytest.mark.gpu
def test_hip_level_deduplication_changes_nothing(monkeypatch):
    """two-layer `recent` TGAT computes every distinct (node, time) entry of level 1 once (k_dedup_*): the result is bit-identical to
    computing all of them (DYGNN_TGAT_DEDUP=0), on a batch where most level-1 entries are duplicates"""
    from dyglib_amd import TGAT, get_neighbor_sampler
    data, nf, ef = syn.make_bipartite_graph(300, 40, 20000, seed=33)
    nf[1:] = np.random.RandomState(6).standard_normal(nf[1:].shape).astype(np.float32) * 0.5
    sampler = get_neighbor_sampler(data, "recent", seed=1, device="cuda:0")
    m = TGAT(nf, ef, sampler, 100, num_layers=2, num_heads=2, dropout=0.1, device="cuda:0")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_tgat_params(10).items()})
    m = m.to("cuda:0").eval()
    idx = np.concatenate([np.arange(5), np.arange(data.num_interactions - 1200, data.num_interactions)])      # incl. nodes without history
    src, dst, t = data.src_node_ids[idx], data.dst_node_ids[idx], data.node_interact_times[idx]
    with torch.no_grad():
        monkeypatch.setenv("DYGNN_TGAT_DEDUP", "0")
        a_s, a_d = m.compute_src_dst_node_temporal_embeddings(src, dst, t, num_neighbors=20)
        monkeypatch.setenv("DYGNN_TGAT_DEDUP", "1")
        b_s, b_d = m.compute_src_dst_node_temporal_embeddings(src, dst, t, num_neighbors=20)
        c_s, c_d = m.compute_src_dst_node_temporal_embeddings(src, dst, t, num_neighbors=20)
    assert torch.equal(a_s, b_s) and torch.equal(a_d, b_d)
    assert torch.equal(b_s, c_s) and torch.equal(b_d, c_d)            # and it is reproducible although the representatives are chosen by a race
    # the number of computed entries = roots + distinct (node, time) pairs of level 1 (roots at their float64 times, neighbours at float32 times)
    total, computed = m.last_level_entries()
    roots, rt = np.concatenate([src, dst]), np.concatenate([t, t])
    nb, _, nt = sampler.get_historical_neighbors(roots, rt, 20)
    ids1 = np.concatenate([roots, nb.reshape(-1)]).astype(np.int64)
    t1 = np.concatenate([rt, nt.reshape(-1).astype(np.float64)])
    distinct = len(set(zip(ids1.tolist(), t1.view(np.int64).tolist())))
    assert total == 2 * len(src) * 22 and computed == 2 * len(src) + distinct and computed < total // 2


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["gemm", "chains"])
def test_hip_other_feature_dims_against_oracle(form, monkeypatch):
    """feature dims that are not multiples of the 16-wide tiles / chunks of either layer form (node 40, edge 24, time 24 -> query dim 64, head dim 32,
    key / value input 88) and k = 7 (idle row slots in the attention): both forms against the oracle"""
    from dyglib_amd import TGAT, get_neighbor_sampler
    data, nf, ef = syn.make_bipartite_graph(300, 50, 12000, seed=41, edge_feat_dim=24)
    nf = np.ascontiguousarray(nf[:, :40])
    nf[1:] = np.random.RandomState(8).standard_normal(nf[1:].shape).astype(np.float32) * 0.5
    params = syn.make_tgat_params(12, node_feat_dim=40, edge_feat_dim=24, time_feat_dim=24, num_layers=2)
    sampler = get_neighbor_sampler(data, "recent", seed=1, device="cuda:0")
    m = TGAT(nf, ef, sampler, 24, num_layers=2, num_heads=2, dropout=0.1, device="cuda:0")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    m = m.to("cuda:0").eval()
    adj = orc.OracleAdjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    idx = np.concatenate([np.arange(3), np.arange(data.num_interactions - 150, data.num_interactions)])      # incl. nodes without history
    src, dst, t = data.src_node_ids[idx], data.dst_node_ids[idx], data.node_interact_times[idx]
    os_, od = torc.tgat_forward(params, nf, ef, adj, src, dst, t, 2, 7, 2)
    if form == "chains":
        monkeypatch.setenv("DYGNN_TGAT_CHAIN", "1")
    with torch.no_grad():
        gs, gd = m.compute_src_dst_node_temporal_embeddings(src, dst, t, num_neighbors=7)
    close(gs.cpu().numpy(), os_.numpy(), f"other dims ({form}) src")
    close(gd.cpu().numpy(), od.numpy(), f"other dims ({form}) dst")


@pytest.mark.gpu
@pytest.mark.parametrize("k,heads", [(24, 2), (12, 4), (24, 4)])
def test_hip_general_attention_kernels_against_oracle(k, heads):
    """Configurations outside the two-waves-per-node attention kernel (k <= 20, <= 2 heads): more than 20 neighbours take k_tgat_attn_lin<0>
    (row slots in LDS), more than two heads k_tgat_attn_lin<20> — the reference accepts any num_neighbors / num_heads
    (models/TGAT.py:14-16, models/modules.py:99-135).  No reference fixture holds these shapes: the oracle is the bar."""
    from dyglib_amd import TGAT, get_neighbor_sampler
    data, nf, ef = syn.make_bipartite_graph(400, 50, 20000, seed=37)
    nf[1:] = np.random.RandomState(5).standard_normal(nf[1:].shape).astype(np.float32) * 0.5
    params = syn.make_tgat_params(14)
    sampler = get_neighbor_sampler(data, "recent", seed=1, device="cuda:0")
    m = TGAT(nf, ef, sampler, 100, num_layers=2, num_heads=heads, dropout=0.1, device="cuda:0")
    m.load_state_dict({kk: torch.from_numpy(v) for kk, v in params.items()})
    m = m.to("cuda:0").eval()
    adj = orc.OracleAdjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    idx = np.concatenate([np.arange(3), np.arange(data.num_interactions - 120, data.num_interactions)])      # incl. nodes without history
    src, dst, t = data.src_node_ids[idx], data.dst_node_ids[idx], data.node_interact_times[idx]
    os_, od = torc.tgat_forward(params, nf, ef, adj, src, dst, t, 2, k, heads)
    with torch.no_grad():
        gs, gd = m.compute_src_dst_node_temporal_embeddings(src, dst, t, num_neighbors=k)
    close(gs.cpu().numpy(), os_.numpy(), f"general attention k={k} heads={heads} src")
    close(gd.cpu().numpy(), od.numpy(), f"general attention k={k} heads={heads} dst")


@pytest.mark.gpu
@pytest.mark.parametrize("B", [200, 37])
def test_hip_step_embeddings_equal_the_two_calls(B):
    """TGAT.compute_step_embeddings (dygnn_tgat_forward_roots: [sources ; destinations ; negative destinations] as one call, every root with
    its own time; an odd number of roots is padded) returns the rows of the positive call and of the negative call bit for bit"""
    from dyglib_amd import TGAT, get_neighbor_sampler
    data, nf, ef = syn.make_bipartite_graph(500, 60, 30000, seed=31)
    nf[1:] = np.random.RandomState(5).standard_normal(nf[1:].shape).astype(np.float32) * 0.5
    sampler = get_neighbor_sampler(data, "recent", seed=1, device="cuda:0")
    m = TGAT(nf, ef, sampler, 100, num_layers=2, num_heads=2, dropout=0.1, device="cuda:0")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_tgat_params(9).items()})
    m = m.to("cuda:0").eval()
    idx = np.arange(data.num_interactions - B, data.num_interactions)
    src, dst, t = data.src_node_ids[idx], data.dst_node_ids[idx], data.node_interact_times[idx]
    neg = syn.random_negative_dst(np.random.RandomState(3), np.unique(data.dst_node_ids), B)
    with torch.no_grad():
        ps, pd = m.compute_src_dst_node_temporal_embeddings(src, dst, t, num_neighbors=20)
        ns, nd = m.compute_src_dst_node_temporal_embeddings(src, neg, t, num_neighbors=20)
        se, de, ne = m.compute_step_embeddings(src, dst, neg, t, num_neighbors=20)
    assert torch.equal(ps, ns)                                          # the negative call's source rows ARE the positive call's
    assert torch.equal(se, ps) and torch.equal(de, pd) and torch.equal(ne, nd)
