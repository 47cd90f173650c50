"""CPU-only checks of the boundary: the C-ABI library builds, loads and exports every symbol
include/dygnn.h declares; the host CSR builder matches the oracle; argument validation maps to the
reference's exception types; the nn.Module mirrors carry the reference's state_dict keys.
No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from dyglib_amd import _build, _capi, synthetic as syn
from dyglib_amd.temporal_csr import TemporalCSR
from oracle import dygformer_oracle as orc
from tests import golden_cases as gc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    _build.build(verbose=False)          # hipcc cross-compiles gfx950 without a GPU
    return _capi.load()


def test_header_symbols_are_exported_and_bound(lib):
    header = open(os.path.join(ROOT, "include", "dygnn.h")).read()
    declared = set(re.findall(r"\b(dygnn_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_capi.SIGNATURES), (declared ^ set(_capi.SIGNATURES))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.dygnn_abi_version() == _capi.ABI_VERSION


def test_struct_layouts_match_header():
    assert C.sizeof(_capi.Csr) == 6 * 8
    assert C.sizeof(_capi.DygformerConfig) == 8 * 4
    assert C.sizeof(_capi.EncoderLayerWeights) == 12 * 8
    assert C.sizeof(_capi.DygformerWeights) == (14 + 12 * _capi.DYGNN_MAX_LAYERS + 2) * 8
    assert C.sizeof(_capi.DygformerTaps) == (5 + _capi.DYGNN_MAX_LAYERS) * 8


@pytest.mark.parametrize("name", ["bip_p2_l64", "gen_p1_l32"])
def test_csr_host_builder_matches_oracle(lib, name):
    d = gc.build_case(name)["data"]
    csr = TemporalCSR.from_interactions(d.src_node_ids, d.dst_node_ids, d.edge_ids, d.node_interact_times)
    adj = orc.OracleAdjacency(d.src_node_ids, d.dst_node_ids, d.edge_ids, d.node_interact_times)
    np.testing.assert_array_equal(csr.indptr, adj.indptr)
    np.testing.assert_array_equal(csr.nbr, adj.nbr)
    np.testing.assert_array_equal(csr.eid, adj.eid)
    np.testing.assert_array_equal(csr.ts, adj.ts)
    assert csr.indptr[1] == 0                                  # row 0 = padding node, empty
    # reference constructor input (adj_list of tuples, utils/utils.py:297-300) gives the same CSR
    adj_list = [[] for _ in range(d.max_node_id + 1)]
    for s, t, e, ts in zip(d.src_node_ids, d.dst_node_ids, d.edge_ids, d.node_interact_times):
        adj_list[s].append((t, e, ts))
        adj_list[t].append((s, e, ts))
    csr2 = TemporalCSR.from_adj_list(adj_list)
    for f in ("indptr", "nbr", "eid", "ts"):
        np.testing.assert_array_equal(getattr(csr, f), getattr(csr2, f))


def test_csr_builder_edge_cases(lib):
    # empty interaction list: only the padding row
    e = np.zeros(0, dtype=np.int64)
    csr = TemporalCSR.from_interactions(e, e, e, np.zeros(0))
    assert csr.num_nodes == 1 and csr.num_entries == 0
    # out-of-range node id -> IndexError like the reference's adj_list[...] indexing
    with pytest.raises(IndexError):
        TemporalCSR.from_interactions(np.array([5]), np.array([1]), np.array([1]), np.array([0.0]), num_nodes=3)
    # self interaction: stored twice under the same node, src entry first
    csr = TemporalCSR.from_interactions(np.array([2, 2]), np.array([2, 1]), np.array([1, 2]), np.array([1.0, 1.0]))
    assert csr.nbr[csr.indptr[2]:csr.indptr[3]].tolist() == [2, 2, 1]
    assert csr.eid[csr.indptr[2]:csr.indptr[3]].tolist() == [1, 1, 2]


def test_argument_validation_without_gpu(lib):
    csr = _capi.Csr(1, 0, 1, None, None, None)    # non-null dummy indptr; never dereferenced (rejected before launch)
    rc = lib.dygnn_sample_recent(C.byref(csr), None, None, 4, 0, None, None, None, None)
    assert rc == -1 and b"greater than 0" in lib.dygnn_last_error()               # utils/utils.py:157
    with pytest.raises(AssertionError):
        _capi.check(rc)
    rc = lib.dygnn_window_lengths(C.byref(csr), None, None, 0, 1, None, None, None, None)
    assert rc == -1 and b"greater than 1" in lib.dygnn_last_error()               # models/DyGFormer.py:209
    cfg = _capi.DygformerConfig(172, 172, 100, 50, 2, 2, 3, 64)                   # 200 % 3 != 0
    assert lib.dygnn_dygformer_packed_bytes(C.byref(cfg)) == 0
    cfg = _capi.DygformerConfig(172, 172, 100, 50, 2, 2, 2, 64)
    assert lib.dygnn_dygformer_packed_bytes(C.byref(cfg)) > 4_000_000                # ~ one copy of the weight matrices
    assert lib.dygnn_dygformer_workspace_bytes(C.byref(cfg), 200) > 0


def test_module_state_dict_matches_reference_keys(lib):
    from dyglib_amd import DyGFormer, MergeLayer, NeighborSampler
    data, nf, ef = syn.make_bipartite_graph(5, 3, 20, seed=0)
    sampler = NeighborSampler(None, "recent", seed=0, csr=TemporalCSR.from_interactions(
        data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times), device="cpu")
    m = DyGFormer(nf, ef, sampler, time_feat_dim=100, channel_embedding_dim=50, patch_size=2, num_layers=2, num_heads=2,
                  dropout=0.1, max_input_sequence_length=64, device="cpu")
    want = syn.dygformer_param_shapes(172, 172, 100, 50, 2, 2)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == want
    assert sum(int(np.prod(s)) for s in got.values()) == 1_052_222
    # golden parameter sets load strictly
    m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_dygformer_params(3, patch_size=2).items()}, strict=True)
    ml = MergeLayer(172, 172, 172, 1)
    assert {k: tuple(v.shape) for k, v in ml.state_dict().items()} == {
        "fc1.weight": (172, 344), "fc1.bias": (172,), "fc2.weight": (1, 172), "fc2.bias": (1,)}
    # time encoder init = reference's 10^-linspace(0,9,100), bias 0 (models/modules.py:20-21)
    np.testing.assert_allclose(m2 := DyGFormer(nf, ef, sampler, 100, 50).time_encoder.w.weight.detach().numpy().ravel(),
                               1 / 10 ** np.linspace(0, 9, 100, dtype=np.float32))
    # no silent CPU path: a CPU-resident model refuses to run
    with pytest.raises(_capi.DygnnError):
        m.eval()
        with torch.no_grad():
            m.compute_src_dst_node_temporal_embeddings(data.src_node_ids[:2], data.dst_node_ids[:2], data.node_interact_times[:2])
    # the training path is HIP too: a CPU-resident model refuses it the same way
    m.train()
    with pytest.raises(_capi.DygnnError):
        m.compute_src_dst_node_temporal_embeddings(data.src_node_ids[:2], data.dst_node_ids[:2], data.node_interact_times[:2])


def test_unsupported_sampling_strategies_raise(lib):
    from dyglib_amd import NeighborSampler
    data, _, _ = syn.make_bipartite_graph(5, 3, 20, seed=0)
    csr = TemporalCSR.from_interactions(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times)
    s = NeighborSampler(None, "uniform", seed=1, csr=csr, device="cpu")
    assert s.seed == 1 and s.sample_neighbor_strategy == "uniform"
    s.reset_random_state()
    with pytest.raises(NotImplementedError):
        s._require_recent()
    with pytest.raises(ValueError):
        NeighborSampler(None, "bogus", csr=csr, device="cpu")._require_recent()
