"""dyglib_amd — MI355X-native (gfx950 HIP) implementation of DyGLib's temporal-neighbour-aggregation
hot path: NeighborSampler.get_historical_neighbors / get_all_first_hop_neighbors and
DyGFormer.compute_src_dst_node_temporal_embeddings, behind the reference's own Python interface.

    from dyglib_amd import DyGFormer, MergeLayer, get_neighbor_sampler      # instead of models.* / utils.utils

The compute lives in dyglib_amd/csrc (C ABI: include/dygnn.h).  There is no CPU fallback.
"""
from .synthetic import InteractionData  # noqa: F401

__all__ = ["DyGFormer", "TGAT", "MemoryModel", "MergeLayer", "TimeEncoder", "NeighborSampler", "get_neighbor_sampler", "TemporalCSR",
           "count_nodes_appearances", "InteractionData", "Data", "get_link_prediction_data",
           "get_link_prediction_metrics", "get_node_classification_metrics", "link_prediction_metrics_device",
           "NegativeEdgeSampler", "get_idx_data_loader", "evaluate_model_link_prediction"]


def __getattr__(name):
    # lazy: importing the package (e.g. for dyglib_amd.synthetic) must not require the HIP library
    if name in ("DyGFormer",):
        from .dygformer import DyGFormer
        return DyGFormer
    if name == "MemoryModel":
        from .memory_model import MemoryModel
        return MemoryModel
    if name == "TGAT":
        from .tgat import TGAT
        return TGAT
    if name in ("MergeLayer", "TimeEncoder"):
        from . import modules
        return getattr(modules, name)
    if name in ("NeighborSampler", "get_neighbor_sampler", "count_nodes_appearances"):
        from . import neighbor_sampler
        return getattr(neighbor_sampler, name)
    if name in ("Data", "get_link_prediction_data"):
        from . import data_loader
        return getattr(data_loader, name)
    if name in ("get_link_prediction_metrics", "get_node_classification_metrics", "link_prediction_metrics_device"):
        from . import metrics
        return getattr(metrics, name)
    if name in ("NegativeEdgeSampler", "get_idx_data_loader", "evaluate_model_link_prediction"):
        from . import evaluate
        return getattr(evaluate, name)
    if name == "TemporalCSR":
        from .temporal_csr import TemporalCSR
        return TemporalCSR
    raise AttributeError(name)
