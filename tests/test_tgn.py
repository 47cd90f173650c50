"""TGN (BASELINE config 5): a chronological run of batches (negative call, then positive call that updates the memory
bank) — oracle vs reference-generated golden vectors (CPU), HIP path through the C ABI vs golden (GPU), including the
final memory table."""
import numpy as np
import pytest
import torch

from dyglib_amd import synthetic as syn
from oracle import dygformer_oracle as orc
from oracle import tgn_oracle as tn
from tests import golden_cases as gc
from tests.parity import close  # plain 1e-4 absolute; observed errors are printed at the end of the run

TOL = 1e-4




@pytest.fixture(scope="module", params=list(gc.TGN_CASES))
def case(request):
    c = gc.build_tgn_case(request.param)
    return request.param, c, gc.load_golden(request.param)


def test_oracle_matches_reference_golden(case):
    name, c, g = case
    d, cfg = c["data"], c["tgn_cfg"]
    adj = orc.OracleAdjacency(d.src_node_ids, d.dst_node_ids, d.edge_ids, d.node_interact_times)
    st = tn.TgnState(c["node_feat"].shape[0], 172)
    for i, b in enumerate(c["tgn_batches"]):
        ns, nd = tn.tgn_forward(c["tgn_params"], c["node_feat"], c["edge_feat"], adj, st, b["src"], b["neg"], b["t"], None, False,
                                cfg["num_layers"], cfg["num_neighbors"], cfg["num_heads"])
        ps, pd = tn.tgn_forward(c["tgn_params"], c["node_feat"], c["edge_feat"], adj, st, b["src"], b["dst"], b["t"], b["eid"], True,
                                cfg["num_layers"], cfg["num_neighbors"], cfg["num_heads"])
        close(nd.numpy(), g[f"b{i}_neg_dst"], f"{name} batch {i} neg dst")
        close(ps.numpy(), g[f"b{i}_pos_src"], f"{name} batch {i} pos src")
        close(pd.numpy(), g[f"b{i}_pos_dst"], f"{name} batch {i} pos dst")
    close(st.M.numpy(), g["final_memory"], name + " memory")
    close(st.U.numpy(), g["final_last_update"], name + " last update")


def test_state_dict_keys_match_reference(case):
    from dyglib_amd import MemoryModel, NeighborSampler
    from dyglib_amd.temporal_csr import TemporalCSR
    name, c, g = case
    d, cfg = c["data"], c["tgn_cfg"]
    sampler = NeighborSampler(None, "recent", seed=0, csr=TemporalCSR.from_interactions(
        d.src_node_ids, d.dst_node_ids, d.edge_ids, d.node_interact_times), device="cpu")
    m = MemoryModel(c["node_feat"], c["edge_feat"], sampler, time_feat_dim=100, model_name="TGN", num_layers=cfg["num_layers"],
                    num_heads=2, dropout=0.1, device="cpu")
    assert sorted(m.state_dict().keys()) == g["state_dict_keys"].tolist()
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == syn.tgn_param_shapes(c["node_feat"].shape[0], num_layers=cfg["num_layers"])
    with pytest.raises(NotImplementedError):
        MemoryModel(c["node_feat"], c["edge_feat"], sampler, 100, model_name="JODIE")
    with pytest.raises(ValueError):
        MemoryModel(c["node_feat"], c["edge_feat"], sampler, 100, model_name="bogus")


@pytest.mark.gpu
def test_hip_matches_golden(case):
    from dyglib_amd import MemoryModel, get_neighbor_sampler
    name, c, g = case
    d, cfg = c["data"], c["tgn_cfg"]
    dev = "cuda:0"
    sampler = get_neighbor_sampler(d, "recent", seed=1, device=dev)
    m = MemoryModel(c["node_feat"], c["edge_feat"], sampler, time_feat_dim=100, model_name="TGN", num_layers=cfg["num_layers"],
                    num_heads=2, dropout=0.1, device=dev)
    sd = m.state_dict()
    sd.update({k: torch.from_numpy(v) for k, v in c["tgn_params"].items()})
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    m.memory_bank.__init_memory_bank__()
    k = cfg["num_neighbors"]
    backup = None
    with torch.no_grad():
        for i, b in enumerate(c["tgn_batches"]):
            if i == 2:
                backup = m.memory_bank.backup_memory_bank()
            ns, nd = m.compute_src_dst_node_temporal_embeddings(b["src"], b["neg"], b["t"], edge_ids=None, edges_are_positive=False, num_neighbors=k)
            ps, pd = m.compute_src_dst_node_temporal_embeddings(b["src"], b["dst"], b["t"], edge_ids=b["eid"], edges_are_positive=True, num_neighbors=k)
            close(ns.cpu().numpy(), g[f"b{i}_neg_src"], f"{name} batch {i} neg src")
            close(nd.cpu().numpy(), g[f"b{i}_neg_dst"], f"{name} batch {i} neg dst")
            close(ps.cpu().numpy(), g[f"b{i}_pos_src"], f"{name} batch {i} pos src")
            close(pd.cpu().numpy(), g[f"b{i}_pos_dst"], f"{name} batch {i} pos dst")
        close(m.memory_bank.node_memories.data.cpu().numpy(), g["final_memory"], name + " memory")
        close(m.memory_bank.node_last_updated_times.data.cpu().numpy(), g["final_last_update"], name + " last update")
        # reload the backup taken before batch 2 and replay: same results (backup/reload round trip, MemoryModel.py:345-366)
        m.memory_bank.reload_memory_bank(backup)
        b = c["tgn_batches"][2]
        ps, pd = m.compute_src_dst_node_temporal_embeddings(b["src"], b["dst"], b["t"], edge_ids=b["eid"], edges_are_positive=True, num_neighbors=k)
        close(ps.cpu().numpy(), g["b2_pos_src"], name + " replay")
        with pytest.raises(AssertionError):
            m.compute_src_dst_node_temporal_embeddings(b["src"], b["dst"], b["t"], edge_ids=None, edges_are_positive=True, num_neighbors=k)


@pytest.mark.gpu
def test_fused_step_equals_negative_then_positive_call():
    """compute_step_embeddings (dygnn_tgn_forward_step: [positives ; negatives] in one call, the first half updates the state) is
    bit-identical, in its rows and in the memory bank it leaves behind, to the reference's negative call followed by the positive call"""
    import torch
    from dyglib_amd import MemoryModel, get_neighbor_sampler
    c = gc.build_tgn_case("tgn_bip_l1_k10")
    cfg, d, dev = c["tgn_cfg"], c["data"], "cuda:0"

    def build():
        sampler = get_neighbor_sampler(d, "recent", seed=1, device=dev)
        m = MemoryModel(c["node_feat"], c["edge_feat"], sampler, time_feat_dim=cfg["time_feat_dim"], model_name="TGN", num_layers=cfg["num_layers"],
                        num_heads=cfg["num_heads"], dropout=0.1, device=dev)
        sd = m.state_dict(); sd.update({k: torch.from_numpy(v) for k, v in c["tgn_params"].items()}); m.load_state_dict(sd)
        m = m.to(dev).eval()
        m.memory_bank.__init_memory_bank__()
        return m

    two, one = build(), build()
    k = cfg["num_neighbors"]
    with torch.no_grad():
        for b in c["tgn_batches"]:
            ns, nd = two.compute_src_dst_node_temporal_embeddings(b["src"], b["neg"], b["t"], edge_ids=None, edges_are_positive=False, num_neighbors=k)
            ps, pd = two.compute_src_dst_node_temporal_embeddings(b["src"], b["dst"], b["t"], edge_ids=b["eid"], edges_are_positive=True, num_neighbors=k)
            fs, fd, fns, fnd = one.compute_step_embeddings(b["src"], b["dst"], b["src"], b["neg"], b["t"], b["eid"], num_neighbors=k)
            for x, y in ((ps, fs), (pd, fd), (ns, fns), (nd, fnd)):
                assert torch.equal(x, y)
    assert torch.equal(two.memory_bank.node_memories, one.memory_bank.node_memories)
    assert torch.equal(two.memory_bank.node_last_updated_times, one.memory_bank.node_last_updated_times)
    assert torch.equal(two.memory_bank.msg, one.memory_bank.msg) and torch.equal(two.memory_bank.has_msg, one.memory_bank.has_msg)


@pytest.mark.gpu
@pytest.mark.parametrize("layers,K,NB", [(1, 10, 12), (2, 6, 5)])
def test_hip_mid_size_batches_against_oracle(layers, K, NB):
    """BASELINE config 5's call shape on a mid-size graph: B = 200, k = 10, 1 layer, 12 chronological batches from interaction 0 with
    the memory carried across them (negative call, then positive call, per batch — through the one-call step), against
    oracle/tgn_oracle.py: every batch's four embedding blocks, and the memory bank + last-update times left behind.  And a two-layer model
    (k = 6): the row-block chains over a de-duplicated level 1 (device-side row count, rows of the level below through the map)."""
    import torch
    from dyglib_amd import MemoryModel, get_neighbor_sampler
    from oracle import dygformer_oracle as orc
    dev, B = "cuda:0", 200
    data, nf, ef = syn.make_bipartite_graph(700, 60, 40_000, seed=11, edge_feat_kind="sparse4", duplicate_time_every=13)
    params = syn.make_tgn_params(5, nf.shape[0], num_layers=layers)
    sampler = get_neighbor_sampler(data, "recent", seed=1, device=dev)
    m = MemoryModel(nf, ef, sampler, 100, model_name="TGN", num_layers=layers, num_heads=2, dropout=0.1, device=dev)
    sd = m.state_dict(); sd.update({k: torch.from_numpy(v) for k, v in params.items()}); m.load_state_dict(sd)
    m = m.to(dev).eval()
    m.memory_bank.__init_memory_bank__()
    adj = orc.OracleAdjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    nft, eft = torch.from_numpy(nf), torch.from_numpy(ef)
    tp = {k: torch.from_numpy(v) for k, v in params.items()}
    st = tn.TgnState(nf.shape[0], 172)
    rs, ud = np.random.RandomState(2), np.unique(data.dst_node_ids)
    with torch.no_grad():
        for i in range(NB):
            sl = slice(i * B, (i + 1) * B)
            s, d, t, e = data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl], data.edge_ids[sl]
            ng = syn.random_negative_dst(rs, ud, B)
            ps, pd, ns, nd = m.compute_step_embeddings(s, d, s, ng, t, e, num_neighbors=K)
            ons, ond = tn.tgn_forward(tp, nft, eft, adj, st, s, ng, t, None, False, layers, K, 2)
            ops, opd = tn.tgn_forward(tp, nft, eft, adj, st, s, d, t, e, True, layers, K, 2)
            for got, want, what in ((ps, ops, "pos src"), (pd, opd, "pos dst"), (ns, ons, "neg src"), (nd, ond, "neg dst")):
                close(got.cpu().numpy(), want.numpy(), f"tgn mid-size batch {i} {what}", label=f"tgn mid-size (B=200, k={K}, {layers} layer(s), {NB} batches) {what}")
    close(m.memory_bank.node_memories.data.cpu().numpy(), st.M.numpy(), f"tgn mid-size memory after {NB} batches ({layers} layer(s))")
    close(m.memory_bank.node_last_updated_times.data.cpu().numpy(), st.U.numpy(), f"tgn mid-size last update after {NB} batches ({layers} layer(s))")


@pytest.mark.gpu
def test_joint_step_equals_the_four_array_form():
    """compute_step_embeddings_joint (the step as ONE batch [positives ; negatives], no per-step concatenation kernels) returns the rows of
    compute_step_embeddings and leaves the same memory bank (MemoryModel.py:87-168 over evaluate_models_utils.py:85-107)."""
    import numpy as np
    import torch
    from dyglib_amd import MemoryModel, get_neighbor_sampler
    c = gc.build_tgn_case("tgn_bip_l1_k10")
    cfg, d, dev = c["tgn_cfg"], c["data"], "cuda:0"

    def build():
        sampler = get_neighbor_sampler(d, "recent", seed=1, device=dev)
        m = MemoryModel(c["node_feat"], c["edge_feat"], sampler, time_feat_dim=cfg["time_feat_dim"], model_name="TGN", num_layers=cfg["num_layers"],
                        num_heads=cfg["num_heads"], dropout=0.1, device=dev)
        sd = m.state_dict(); sd.update({k: torch.from_numpy(v) for k, v in c["tgn_params"].items()}); m.load_state_dict(sd)
        m = m.to(dev).eval()
        m.memory_bank.__init_memory_bank__()
        return m

    four, joint = build(), build()
    k = cfg["num_neighbors"]
    with torch.no_grad():
        for b in c["tgn_batches"]:
            n = len(b["src"])
            ps, pd, ns, nd = four.compute_step_embeddings(b["src"], b["dst"], b["src"], b["neg"], b["t"], b["eid"], num_neighbors=k)
            js, jd = joint.compute_step_embeddings_joint(np.concatenate([b["src"], b["src"]]), np.concatenate([b["dst"], b["neg"]]), np.concatenate([b["t"], b["t"]]),
                                                         b["eid"], n, num_neighbors=k)
            assert torch.equal(js[:n], ps) and torch.equal(jd[:n], pd) and torch.equal(js[n:], ns) and torch.equal(jd[n:], nd)
    assert torch.equal(four.memory_bank.node_memories, joint.memory_bank.node_memories)
    assert torch.equal(four.memory_bank.msg, joint.memory_bank.msg) and torch.equal(four.memory_bank.has_msg, joint.memory_bank.has_msg)


@pytest.mark.gpu
def test_rows_and_state_do_not_depend_on_the_workgroup_shapes(monkeypatch):
    """The row-block kernels of a TGN call (tgat_chain.hip) pick rows per workgroup from the level size, and the GRU kernel splits the
    memory dims into slices: rows and the memory bank left behind are bit-identical for every choice (DYGNN_CHAIN_MT / DYGNN_GRU_MT /
    DYGNN_GRU_SLICES force one), and the layer's product-by-product GEMM form (DYGNN_TGAT_CHAIN=0) agrees to rounding."""
    import numpy as np
    import torch
    from dyglib_amd import MemoryModel, get_neighbor_sampler
    c = gc.build_tgn_case("tgn_bip_l1_k10")
    cfg, d, dev = c["tgn_cfg"], c["data"], "cuda:0"

    def run(env):
        for k_, v in env.items():
            monkeypatch.setenv(k_, v)
        sampler = get_neighbor_sampler(d, "recent", seed=1, device=dev)
        m = MemoryModel(c["node_feat"], c["edge_feat"], sampler, time_feat_dim=cfg["time_feat_dim"], model_name="TGN", num_layers=cfg["num_layers"],
                        num_heads=cfg["num_heads"], dropout=0.1, device=dev)
        sd = m.state_dict(); sd.update({k: torch.from_numpy(v) for k, v in c["tgn_params"].items()}); m.load_state_dict(sd)
        m = m.to(dev).eval()
        m.memory_bank.__init_memory_bank__()
        outs = []
        with torch.no_grad():
            for b in c["tgn_batches"]:
                outs.append(torch.cat(m.compute_step_embeddings_joint(np.concatenate([b["src"], b["src"]]), np.concatenate([b["dst"], b["neg"]]),
                                                                      np.concatenate([b["t"], b["t"]]), b["eid"], len(b["src"]), num_neighbors=cfg["num_neighbors"])))
        for k_ in env:
            monkeypatch.delenv(k_)
        return torch.cat(outs), m.memory_bank.node_memories.clone(), m.memory_bank.msg.clone()

    base = run({})
    for env in ({"DYGNN_CHAIN_MT": "2", "DYGNN_GRU_MT": "2", "DYGNN_GRU_SLICES": "1"}, {"DYGNN_CHAIN_MT": "4", "DYGNN_GRU_MT": "4", "DYGNN_GRU_SLICES": "3"},
                {"DYGNN_CHAIN_MT": "8", "DYGNN_GRU_MT": "1", "DYGNN_GRU_SLICES": "6"}):
        other = run(env)
        assert all(torch.equal(x, y) for x, y in zip(base, other)), env
    gemm = run({"DYGNN_TGAT_CHAIN": "0"})
    close(base[0].cpu().numpy(), gemm[0].cpu().numpy(), "tgn chain vs GEMM form: rows")
    close(base[1].cpu().numpy(), gemm[1].cpu().numpy(), "tgn chain vs GEMM form: memory")


@pytest.mark.gpu
def test_hip_other_feature_dims_against_oracle():
    """feature dims that are not multiples of the 16-wide tiles / chunks of the row-block kernels (node = memory 40, edge 24, time 24: message
    128, GRU gates 120 = 7.5 tiles split over the slices), two layers, k = 7: sequential batches against the oracle incl. the memory bank"""
    import torch
    from dyglib_amd import MemoryModel, get_neighbor_sampler
    from oracle import dygformer_oracle as orc
    dev, B, K, NB, Fn, Fe, Ft, layers = "cuda:0", 120, 7, 6, 40, 24, 24, 2
    data, nf, ef = syn.make_bipartite_graph(300, 40, 12_000, seed=43, edge_feat_dim=Fe, duplicate_time_every=11)
    nf = np.ascontiguousarray(nf[:, :Fn])
    nf[1:] = np.random.RandomState(9).standard_normal(nf[1:].shape).astype(np.float32) * 0.5
    rs = np.random.RandomState(21)
    tg = syn.make_tgat_params(13, node_feat_dim=Fn, edge_feat_dim=Fe, time_feat_dim=Ft, num_layers=layers)
    params = {("embedding_module." + k if not k.startswith("time_encoder") else k): v for k, v in tg.items()}
    for k in ("time_encoder.w.weight", "time_encoder.w.bias"):
        params["embedding_module." + k] = params[k]
    Dm, b = 2 * Fn + Ft + Fe, 1.0 / np.sqrt(Fn)
    params["memory_updater.memory_updater.weight_ih"] = rs.uniform(-b, b, (3 * Fn, Dm)).astype(np.float32)
    params["memory_updater.memory_updater.weight_hh"] = rs.uniform(-b, b, (3 * Fn, Fn)).astype(np.float32)
    params["memory_updater.memory_updater.bias_ih"] = rs.uniform(-b, b, (3 * Fn,)).astype(np.float32)
    params["memory_updater.memory_updater.bias_hh"] = rs.uniform(-b, b, (3 * Fn,)).astype(np.float32)
    sampler = get_neighbor_sampler(data, "recent", seed=1, device=dev)
    m = MemoryModel(nf, ef, sampler, Ft, model_name="TGN", num_layers=layers, num_heads=2, dropout=0.1, device=dev)
    sd = m.state_dict(); sd.update({k: torch.from_numpy(v) for k, v in params.items()}); m.load_state_dict(sd)
    m = m.to(dev).eval()
    m.memory_bank.__init_memory_bank__()
    adj = orc.OracleAdjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    nft, eft = torch.from_numpy(nf), torch.from_numpy(ef)
    tp = {k: torch.from_numpy(v) for k, v in params.items()}
    st = tn.TgnState(nf.shape[0], Fn)
    ud = np.unique(data.dst_node_ids)
    with torch.no_grad():
        for i in range(NB):
            sl = slice(i * B, (i + 1) * B)
            s, d, t, e = data.src_node_ids[sl], data.dst_node_ids[sl], data.node_interact_times[sl], data.edge_ids[sl]
            ng = syn.random_negative_dst(rs, ud, B)
            ps, pd, ns, nd = m.compute_step_embeddings(s, d, s, ng, t, e, num_neighbors=K)
            ons, ond = tn.tgn_forward(tp, nft, eft, adj, st, s, ng, t, None, False, layers, K, 2)
            ops, opd = tn.tgn_forward(tp, nft, eft, adj, st, s, d, t, e, True, layers, K, 2)
            for got, want, what in ((ps, ops, "pos src"), (pd, opd, "pos dst"), (ns, ons, "neg src"), (nd, ond, "neg dst")):
                close(got.cpu().numpy(), want.numpy(), f"tgn other dims batch {i} {what}", label=f"tgn other dims {what}")
    close(m.memory_bank.node_memories.data.cpu().numpy(), st.M.numpy(), "tgn other dims memory")
    close(m.memory_bank.node_last_updated_times.data.cpu().numpy(), st.U.numpy(), "tgn other dims last update")
