"""Dataset files of the reference (`utils/DataLoader.py:46-168`): `processed_data/<name>/ml_<name>.csv` with columns
(index), u, i, ts, label, idx; `ml_<name>.npy` = edge features [E+1, F_e]; `ml_<name>_node.npy` = node features
[N+1, F_n]; features zero-padded to 172 columns; chronological 70/15/15 split by timestamp quantiles and the
inductive new-node split seeded with 2020.  Same function name, arguments and return tuple as the reference, so
`train_link_prediction.py:33-34` can import it from here; the graph then goes straight into the C++ CSR builder
(`get_neighbor_sampler`) instead of the reference's per-node Python lists.
"""
from __future__ import annotations

import os
import random
import warnings
from typing import Tuple

import numpy as np

NODE_FEAT_DIM = EDGE_FEAT_DIM = 172          # utils/DataLoader.py:82


class Data:
    """utils/DataLoader.py:46-64."""

    def __init__(self, src_node_ids: np.ndarray, dst_node_ids: np.ndarray, node_interact_times: np.ndarray, edge_ids: np.ndarray,
                 labels: np.ndarray):
        self.src_node_ids = src_node_ids
        self.dst_node_ids = dst_node_ids
        self.node_interact_times = node_interact_times
        self.edge_ids = edge_ids
        self.labels = labels
        self.num_interactions = len(src_node_ids)
        self.unique_node_ids = set(src_node_ids) | set(dst_node_ids)
        self.num_unique_nodes = len(self.unique_node_ids)


def _read_graph_csv(path: str):
    """-> u, i, ts, label, idx columns (the unnamed first column is the row index written by pandas)."""
    try:
        import pandas as pd
        df = pd.read_csv(path)
        return df.u.values, df.i.values, df.ts.values, df.label.values, df.idx.values
    except ImportError:                                        # plain numpy fallback
        with open(path) as f:
            header = f.readline().strip().split(",")
        cols = {name: k for k, name in enumerate(header)}
        arr = np.loadtxt(path, delimiter=",", skiprows=1, ndmin=2)
        return (arr[:, cols["u"]].astype(np.int64), arr[:, cols["i"]].astype(np.int64), arr[:, cols["ts"]],
                arr[:, cols["label"]], arr[:, cols["idx"]].astype(np.int64))


def _pad_features(x: np.ndarray, dim: int, what: str, dataset_name: str) -> np.ndarray:
    assert dim >= x.shape[1], f"{what} feature dimension in dataset {dataset_name} is bigger than {dim}!"      # utils/DataLoader.py:83-84
    if x.shape[1] < dim:
        x = np.concatenate([x, np.zeros((x.shape[0], dim - x.shape[1]))], axis=1)                             # utils/DataLoader.py:86-91
    return x


def load_dataset_files(dataset_name: str, root: str = "./processed_data"):
    d = os.path.join(root, dataset_name)
    u, i, ts, label, idx = _read_graph_csv(os.path.join(d, f"ml_{dataset_name}.csv"))
    edge_raw_features = np.load(os.path.join(d, f"ml_{dataset_name}.npy"), allow_pickle=False)
    node_raw_features = np.load(os.path.join(d, f"ml_{dataset_name}_node.npy"), allow_pickle=False)
    node_raw_features = _pad_features(node_raw_features, NODE_FEAT_DIM, "Node", dataset_name)
    edge_raw_features = _pad_features(edge_raw_features, EDGE_FEAT_DIM, "Edge", dataset_name)
    assert NODE_FEAT_DIM == node_raw_features.shape[1] and EDGE_FEAT_DIM == edge_raw_features.shape[1], \
        "Unaligned feature dimensions after feature padding!"
    return (node_raw_features, edge_raw_features, u.astype(np.longlong), i.astype(np.longlong), ts.astype(np.float64),
            idx.astype(np.longlong), label)


def get_link_prediction_data(dataset_name: str, val_ratio: float, test_ratio: float, root: str = "./processed_data") -> Tuple:
    """utils/DataLoader.py:67-168: node_raw_features, edge_raw_features, full_data, train_data, val_data, test_data,
    new_node_val_data, new_node_test_data."""
    node_raw_features, edge_raw_features, src, dst, t, eid, labels = load_dataset_files(dataset_name, root)
    val_time, test_time = list(np.quantile(t, [(1 - val_ratio - test_ratio), (1 - test_ratio)]))             # utils/DataLoader.py:95

    full_data = Data(src, dst, t, eid, labels)

    random.seed(2020)                                                                                         # utils/DataLoader.py:106
    node_set = set(src) | set(dst)
    num_total_unique_node_ids = len(node_set)
    # nodes seen after the validation time; the sample below must see the SAME set iteration order as the reference
    test_node_set = set(src[t > val_time]).union(set(dst[t > val_time]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        new_test_node_set = set(random.sample(tuple(test_node_set), int(0.1 * num_total_unique_node_ids)))    # utils/DataLoader.py:115
    new_nodes = np.fromiter(new_test_node_set, dtype=np.int64, count=len(new_test_node_set))

    new_test_source_mask = np.isin(src, new_nodes)
    new_test_destination_mask = np.isin(dst, new_nodes)
    observed_edges_mask = np.logical_and(~new_test_source_mask, ~new_test_destination_mask)
    train_mask = np.logical_and(t <= val_time, observed_edges_mask)

    def pick(m):
        return Data(src[m], dst[m], t[m], eid[m], labels[m])

    train_data = pick(train_mask)
    train_node_set = set(train_data.src_node_ids).union(train_data.dst_node_ids)
    assert len(train_node_set & new_test_node_set) == 0
    new_node_set = node_set - train_node_set
    new_node_arr = np.fromiter(new_node_set, dtype=np.int64, count=len(new_node_set))

    val_mask = np.logical_and(t <= test_time, t > val_time)
    test_mask = t > test_time
    edge_contains_new_node_mask = np.logical_or(np.isin(src, new_node_arr), np.isin(dst, new_node_arr))
    new_node_val_mask = np.logical_and(val_mask, edge_contains_new_node_mask)
    new_node_test_mask = np.logical_and(test_mask, edge_contains_new_node_mask)

    return (node_raw_features, edge_raw_features, full_data, train_data, pick(val_mask), pick(test_mask),
            pick(new_node_val_mask), pick(new_node_test_mask))
