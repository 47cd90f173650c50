"""GPU parity of DyGFormer.compute_src_dst_node_temporal_embeddings through the C ABI.

Tolerance: BASELINE.json's north_star asks for fp32 embeddings within 1e-4 of the reference CPU
path.  Embeddings and link probabilities — the outputs of the path — are held to a plain absolute
1e-4 (tests/parity.py), also on the L=512 stress case whose embeddings reach magnitude ~20.  The
internal taps (encoder input, per-layer residual stream) reach magnitude ~140 there, where one
float32 ulp is 1.5e-5: they use 1e-4 * max(1, max|ref|).  The largest observed error per label is
printed at the end of the run."""
import numpy as np
import pytest
import torch

from dyglib_amd import synthetic as syn
from oracle import dygformer_oracle as orc
from tests import golden_cases as gc
from tests.parity import close, close_scaled  # embeddings / probabilities: plain 1e-4 absolute; internal taps (magnitude up to ~140): scaled

pytestmark = pytest.mark.gpu

TOL = 1e-4
IMPLS = {"generic": 1, "fused3": 3, "auto": 0}
GENERIC_UNSUPPORTED = {"bip_p64_l2048"}                   # impl 1: 4096 window positions do not fit its LDS staging




def build_model(c, device="cuda:0"):
    from dyglib_amd import DyGFormer, MergeLayer, get_neighbor_sampler
    cfg = c["cfg"]
    sampler = get_neighbor_sampler(c["data"], "recent", seed=1, device=device)
    model = DyGFormer(c["node_feat"], c["edge_feat"], sampler, time_feat_dim=cfg["time_feat_dim"],
                      channel_embedding_dim=cfg["channel_embedding_dim"], patch_size=cfg["patch_size"],
                      num_layers=cfg["num_layers"], num_heads=cfg["num_heads"], dropout=0.1,
                      max_input_sequence_length=cfg["max_input_sequence_length"], device=device)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in c["params"].items()}, strict=True)
    merge = MergeLayer(172, 172, 172, 1)
    merge.load_state_dict({k: torch.from_numpy(v) for k, v in c["mparams"].items()}, strict=True)
    return model.to(device).eval(), merge.to(device).eval()


@pytest.fixture(scope="module", params=list(gc.CASES))
def case(request):
    c = gc.build_case(request.param)
    g = gc.load_golden(request.param)
    model, merge = build_model(c)
    return request.param, c, g, model, merge


@pytest.mark.parametrize("impl", list(IMPLS))
def test_forward_matches_golden(case, impl):
    name, c, g, model, merge = case
    model.impl = IMPLS[impl]
    if impl == "generic" and name in GENERIC_UNSUPPORTED:
        with pytest.raises(NotImplementedError):          # DYGNN_E_UNSUPPORTED, never a silent fallback
            with torch.no_grad():
                model.compute_src_dst_node_temporal_embeddings(c["src"], c["dst"], c["times"])
        return
    taps = {}
    with torch.no_grad():
        se, de = model.compute_src_dst_node_temporal_embeddings(c["src"], c["dst"], c["times"], _taps=taps)
        nse, nde = model.compute_src_dst_node_temporal_embeddings(c["src"], c["neg_dst"], c["times"])
        pos = merge.link_probabilities(se, de)
        neg = merge.link_probabilities(nse, nde)
        pos_ref_api = merge(se, de).squeeze(-1).sigmoid()
    torch.cuda.synchronize()
    assert se.dtype == torch.float32 and se.shape == (len(c["src"]), 172) and se.is_cuda
    S_s, S_d = g["src_pad_ids"].shape[1], g["dst_pad_ids"].shape[1]
    assert taps["seq_lens"].cpu().tolist() == [S_s, S_d]
    P = c["cfg"]["patch_size"]
    T = (S_s + S_d) // P
    R = gc.TAP_ROWS
    close_scaled(taps["encoder_input"][:R, :T].cpu().numpy(), g["encoder_input_rows"], f"{name}/{impl} encoder input (internal tap, scaled bar)")
    for l in range(2):
        close_scaled(taps["layer_outputs"][l][:R, :T].cpu().numpy(), g[f"layer{l}_rows"], f"{name}/{impl} layer {l} (internal tap, scaled bar)")
    close(se.cpu().numpy(), g["src_emb"], f"{name}/{impl} src emb")
    close(de.cpu().numpy(), g["dst_emb"], f"{name}/{impl} dst emb")
    close(nse.cpu().numpy(), g["neg_src_emb"], f"{name}/{impl} neg src emb")
    close(nde.cpu().numpy(), g["neg_dst_emb"], f"{name}/{impl} neg dst emb")
    close(pos.cpu().numpy(), g["pos_prob"], f"{name}/{impl} pos prob")
    close(neg.cpu().numpy(), g["neg_prob"], f"{name}/{impl} neg prob")
    close(pos.cpu().numpy(), pos_ref_api.cpu().numpy(), "fused head vs MergeLayer.forward")


def test_device_resident_inputs_and_determinism(case):
    name, c, g, model, merge = case
    model.impl = 0
    dev = "cuda:0"
    src = torch.from_numpy(c["src"]).to(dev)
    dst = torch.from_numpy(c["dst"]).to(dev)
    t = torch.from_numpy(c["times"]).to(dev)
    with torch.no_grad():
        a1, b1 = model.compute_src_dst_node_temporal_embeddings(src, dst, t)
        a2, b2 = model.compute_src_dst_node_temporal_embeddings(c["src"], c["dst"], c["times"])
    assert torch.equal(a1, a2) and torch.equal(b1, b2)          # bitwise reproducible, host or device inputs
    close(a1.cpu().numpy(), g["src_emb"], name)


def test_batch_dependence_only_through_seq_lens(case):
    """Appendix B-7: a row's output depends on the rest of the batch only through S_src/S_dst.  Rows
    evaluated alone with the SAME padded lengths must reproduce the batched rows bit-for-bit; here:
    permuting the batch permutes the outputs."""
    name, c, g, model, merge = case
    model.impl = 0
    perm = np.random.RandomState(0).permutation(len(c["src"]))
    with torch.no_grad():
        a, b = model.compute_src_dst_node_temporal_embeddings(c["src"], c["dst"], c["times"])
        ap, bp = model.compute_src_dst_node_temporal_embeddings(c["src"][perm], c["dst"][perm], c["times"][perm])
    assert torch.equal(a[perm], ap) and torch.equal(b[perm], bp)


def test_weight_update_triggers_repack():
    c = gc.build_case("gen_p1_l32")
    model, _ = build_model(c)
    with torch.no_grad():
        a, _ = model.compute_src_dst_node_temporal_embeddings(c["src"], c["dst"], c["times"])
        model.output_layer.bias.add_(1.0)                       # in-place update, like an optimizer step
        b, _ = model.compute_src_dst_node_temporal_embeddings(c["src"], c["dst"], c["times"])
        model.transformers[1].linear_layers[1].weight.mul_(0.5)
        d, _ = model.compute_src_dst_node_temporal_embeddings(c["src"], c["dst"], c["times"])
    np.testing.assert_allclose((b - a).cpu().numpy(), 1.0, atol=1e-5)
    assert not torch.allclose(d, b)


def test_wikipedia_scale_batch_against_oracle():
    """BASELINE config 1/2 shape at full batch (B=200, L=64, P=2) on a mid-size graph, vs the oracle."""
    from dyglib_amd import DyGFormer, get_neighbor_sampler
    data, nf, ef = syn.make_bipartite_graph(600, 80, 20000, seed=21)
    params = syn.make_dygformer_params(7, patch_size=2)
    sampler = get_neighbor_sampler(data, "recent", seed=1, device="cuda:0")
    model = DyGFormer(nf, ef, sampler, 100, 50, patch_size=2, num_layers=2, num_heads=2, dropout=0.1,
                      max_input_sequence_length=64, device="cuda:0")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    model = model.to("cuda:0").eval()
    adj = orc.OracleAdjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    idx = np.arange(data.num_interactions - 200, data.num_interactions)
    src, dst, t = data.src_node_ids[idx], data.dst_node_ids[idx], data.node_interact_times[idx]
    with torch.no_grad():
        os_, od = orc.dygformer_forward(params, nf, ef, adj, src, dst, t, 2, 64)
        for impl in (1, 3):
            model.impl = impl
            gs, gd = model.compute_src_dst_node_temporal_embeddings(src, dst, t)
            close(gs.cpu().numpy(), os_.numpy(), f"impl {impl} src")
            close(gd.cpu().numpy(), od.numpy(), f"impl {impl} dst")


@pytest.mark.parametrize("L,P,paired", [(256, 8, False), (256, 8, True), (256, 4, False)])
def test_long_windows_with_the_hash_table_counts_against_oracle(L, P, paired):
    """Windows of >= 512 positions per pair count their co-occurrences through the LDS hash table of the fused kernel (round 3).  The reference
    fixtures reach it with one pair per workgroup only (L = 512 / P = 8); here: L = 256 / P = 8 = 64 tokens per pair = TWO pairs per workgroup, each
    with its own table (also as the positive / negative pair of one edge), and L = 256 / P = 4 = 128 tokens (one pair) — hub items with many repeated
    neighbours (counts far above 1), empty histories, duplicate timestamps.  No reference fixture holds these shapes: the oracle is the bar."""
    from dyglib_amd import DyGFormer, get_neighbor_sampler, count_nodes_appearances
    data, nf, ef = syn.make_bipartite_graph(40, 6, 9000, seed=23, duplicate_time_every=5)
    nf[1:] = np.random.RandomState(2).standard_normal(nf[1:].shape).astype(np.float32) * 0.3
    params = syn.make_dygformer_params(9, patch_size=P)
    sampler = get_neighbor_sampler(data, "recent", seed=1, device="cuda:0")
    model = DyGFormer(nf, ef, sampler, 100, 50, patch_size=P, num_layers=2, num_heads=2, dropout=0.1, max_input_sequence_length=L, device="cuda:0")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    model = model.to("cuda:0").eval()
    adj = orc.OracleAdjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    idx = np.concatenate([np.arange(4), np.arange(data.num_interactions - 44, data.num_interactions)])       # 48 pairs, incl. first interactions
    src, dst, t = data.src_node_ids[idx], data.dst_node_ids[idx], data.node_interact_times[idx]
    neg = syn.random_negative_dst(np.random.RandomState(4), np.unique(data.dst_node_ids), len(idx))
    with torch.no_grad():
        os_, od = orc.dygformer_forward(params, nf, ef, adj, src, dst, t, P, L)
        ons, ond = orc.dygformer_forward(params, nf, ef, adj, src, neg, t, P, L)
        if paired:       # [positives ; negatives] as one launch, pair i and pair i + B in one workgroup
            s2, d2 = model.compute_src_dst_node_temporal_embeddings_many(np.stack([src, src]), np.stack([dst, neg]), np.stack([t, t]), pos_neg_halves=True)
            gs, gd, gns, gnd = s2[0], d2[0], s2[1], d2[1]
        else:
            gs, gd = model.compute_src_dst_node_temporal_embeddings(src, dst, t)
            gns, gnd = model.compute_src_dst_node_temporal_embeddings(src, neg, t)
    for got, want, tag in ((gs, os_, "src"), (gd, od, "dst"), (gns, ons, "neg src"), (gnd, ond, "neg dst")):
        close(got.cpu().numpy(), want.numpy(), f"L={L} P={P} paired={paired} {tag} emb", label=f"long windows, hash-table counts (L={L}, P={P})")
    # the windows really are long and repetitive: some count exceeds 50
    ids_s, _, _ = sampler.padded_windows(src, t, P, L)
    ids_d, _, _ = sampler.padded_windows(dst, t, P, L)
    cs, cd = count_nodes_appearances(ids_s, ids_d, device="cuda:0")
    assert ids_s.shape[1] + ids_d.shape[1] >= 256 and float(cs.max()) > 50


@pytest.mark.parametrize("name", ["bip_p2_l64", "hub_p4_l48", "gen_p1_l32", "bip_p8_l512"])
def test_positive_and_negative_call_in_one_workgroup_match_separate_calls(name):
    """SURVEY §8f-4 for DyGFormer: [positive calls ; negative calls] with pos_neg_halves=True puts both pairs of an edge in one workgroup
    and projects the shared source side once.  Every row must equal the separate reference calls BIT FOR BIT — for full source tiles
    (the shared path), for source lengths that are not a multiple of 16 tokens (plain path), for a 128-token shape (one pair per
    workgroup: nothing shared) and for "negatives" whose source or time differs (plain path, per pair)."""
    c = gc.build_case(name)
    model, _ = build_model(c)
    d = c["data"]
    E = d.num_interactions
    B = min(24, len(c["src"]))
    rows = [np.arange(E - 2 * B, E - B), np.arange(E - B, E)]          # two positive calls (long histories)
    rs = np.random.RandomState(5)
    src = np.stack([d.src_node_ids[r] for r in rows] * 2)                # negatives: the same sources ...
    t = np.stack([d.node_interact_times[r] for r in rows] * 2)           # ... and times ...
    neg = [rs.choice(np.unique(d.dst_node_ids), size=B) for _ in rows]   # ... other destinations
    dst = np.stack([d.dst_node_ids[r] for r in rows] + neg)
    # a few "negatives" that are NOT the negative of their partner: another source / another time
    src[2, 1], t[3, 2] = src[2, 0], t[3, 0] - 1.0
    model.impl = 3
    with torch.no_grad():
        ps, pd = model.compute_src_dst_node_temporal_embeddings_many(src, dst, t, pos_neg_halves=True)
        qs, qd = model.compute_src_dst_node_temporal_embeddings_many(src, dst, t)
        for i in range(4):
            s1, d1 = model.compute_src_dst_node_temporal_embeddings(src[i], dst[i], t[i])
            assert torch.equal(ps[i], s1) and torch.equal(pd[i], d1), (name, i)
            assert torch.equal(qs[i], s1) and torch.equal(qd[i], d1), (name, i)
    with pytest.raises(AssertionError), torch.no_grad():
        model.compute_src_dst_node_temporal_embeddings_many(src[:3], dst[:3], t[:3], pos_neg_halves=True)


def test_many_calls_in_one_launch_match_separate_calls():
    """group_size: several independent reference calls as ONE grid.  Every group keeps its own padded lengths, so the
    result must equal the separate calls BIT FOR BIT, including groups whose S_src/S_dst differ."""
    c = gc.build_case("hub_p4_l48")
    model, _ = build_model(c)
    d = c["data"]
    E = d.num_interactions
    B = 12
    # group 0: the first interactions (short histories, small S), group 1/2: late ones (long dst histories)
    rows = [np.arange(0, B), np.arange(E - 2 * B, E - B), np.arange(E - B, E)]
    src = np.stack([d.src_node_ids[r] for r in rows])
    dst = np.stack([d.dst_node_ids[r] for r in rows])
    t = np.stack([d.node_interact_times[r] for r in rows])
    for impl in (1, 3):
        model.impl = impl
        with torch.no_grad():
            many_s, many_d = model.compute_src_dst_node_temporal_embeddings_many(src, dst, t)
            lens = []
            for i in range(len(rows)):
                taps = {}
                s1, d1 = model.compute_src_dst_node_temporal_embeddings(src[i], dst[i], t[i], _taps=taps)
                lens.append(tuple(taps["seq_lens"].cpu().tolist()))
                assert torch.equal(many_s[i], s1) and torch.equal(many_d[i], d1), (impl, i)
        assert len(set(lens)) > 1, lens          # the groups really were padded to different lengths


def test_large_timestamps_take_the_libm_cosine_path():
    """Time differences of ~1e9 s (UNIX-epoch style timestamps): w*dt exceeds the 3e7 rad range of the fast range reduction
    for the largest frequencies, so those lanes fall back to libm's cosf — results must still match the oracle."""
    from dyglib_amd import DyGFormer, get_neighbor_sampler
    data, nf, ef = syn.make_bipartite_graph(50, 10, 2500, seed=31)
    data.node_interact_times = np.ascontiguousarray(data.node_interact_times * 1000.0)      # up to 2.7e9
    params = syn.make_dygformer_params(9, patch_size=2)
    sampler = get_neighbor_sampler(data, "recent", seed=1, device="cuda:0")
    model = DyGFormer(nf, ef, sampler, 100, 50, patch_size=2, num_layers=2, num_heads=2, dropout=0.1,
                      max_input_sequence_length=64, device="cuda:0")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    model = model.to("cuda:0").eval()
    adj = orc.OracleAdjacency(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    idx = np.arange(data.num_interactions - 40, data.num_interactions)
    src, dst, t = data.src_node_ids[idx], data.dst_node_ids[idx], data.node_interact_times[idx]
    with torch.no_grad():
        os_, od = orc.dygformer_forward(params, nf, ef, adj, src, dst, t, 2, 64)
        for impl in (1, 3):
            model.impl = impl
            gs, gd = model.compute_src_dst_node_temporal_embeddings(src, dst, t)
            close(gs.cpu().numpy(), os_.numpy(), f"large t, impl {impl} src")
            close(gd.cpu().numpy(), od.numpy(), f"large t, impl {impl} dst")
