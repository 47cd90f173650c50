"""dyglib_amd.data_loader.get_link_prediction_data against the reference's own loader (utils/DataLoader.py:67-168) run
on the same synthetic files in the reference's on-disk format (fixture tests/golden/loader_toy.npz, written by
oracle/make_golden.py); the inductive split depends on `random.sample` over a Python set seeded with 2020."""
import numpy as np

from tests import golden_cases as gc


def test_splits_match_reference_loader(tmp_path):
    from dyglib_amd.data_loader import get_link_prediction_data
    name = gc.write_dataset_files(str(tmp_path))
    g = gc.load_golden("loader_toy")
    r = get_link_prediction_data(name, gc.LOADER_CASE["val_ratio"], gc.LOADER_CASE["test_ratio"],
                                 root=str(tmp_path / "processed_data"))
    assert tuple(r[0].shape) == tuple(g["node_feat_shape"]) and tuple(r[1].shape) == tuple(g["edge_feat_shape"])
    assert str(r[0].dtype) == str(g["node_feat_dtype"]) and str(r[1].dtype) == str(g["edge_feat_dtype"])
    assert r[1].shape[1] == 172 and float(r[1].sum()) == float(g["edge_feat_sum"])     # zero padding to 172 columns
    for tag, d in zip(("full", "train", "val", "test", "new_node_val", "new_node_test"), r[2:]):
        assert np.array_equal(d.edge_ids, g[f"{tag}_edge_ids"]), tag
        assert d.num_unique_nodes == int(g[f"{tag}_num_unique_nodes"]), tag
        assert d.src_node_ids.dtype == np.longlong and d.node_interact_times.dtype == np.float64


def test_loaded_graph_feeds_the_csr_builder(tmp_path):
    from dyglib_amd.data_loader import get_link_prediction_data
    from dyglib_amd.temporal_csr import TemporalCSR
    name = gc.write_dataset_files(str(tmp_path))
    full = get_link_prediction_data(name, 0.15, 0.15, root=str(tmp_path / "processed_data"))[2]
    csr = TemporalCSR.from_interactions(full.src_node_ids, full.dst_node_ids, full.edge_ids, full.node_interact_times)
    assert csr.num_entries == 2 * full.num_interactions and csr.indptr[1] == 0        # row 0 = padding node
    for v in (1, 5, 61):
        row = csr.ts[csr.indptr[v]:csr.indptr[v + 1]]
        assert np.all(np.diff(row) >= 0)
