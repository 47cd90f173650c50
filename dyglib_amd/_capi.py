"""ctypes binding of libdygnn_hip.so (include/dygnn.h).  This is the stub INTEGRATION.md shows.

There is NO fallback: if the library cannot be loaded the import of the product modules raises.
`import torch` happens before the library is loaded so that the process-wide HIP runtime is the
one PyTorch-ROCm ships (same SONAME libamdhip64.so.7): streams and device pointers handed over
by torch are then valid inside the library.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch  # noqa: F401  (must be imported first: see module docstring)

from . import _build

DYGNN_MAX_LAYERS = 8
ABI_VERSION = 14

c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)


class DygnnError(RuntimeError):
    pass


class Csr(C.Structure):
    _fields_ = [("num_nodes", C.c_int64), ("num_entries", C.c_int64), ("indptr", C.c_void_p), ("nbr", C.c_void_p),
                ("eid", C.c_void_p), ("ts", C.c_void_p)]


class DygformerConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("node_feat_dim", "edge_feat_dim", "time_feat_dim", "channel_embedding_dim",
                                         "patch_size", "num_layers", "num_heads", "max_input_sequence_length")]


class EncoderLayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("in_proj_weight", "in_proj_bias", "out_proj_weight", "out_proj_bias",
                                          "ffn0_weight", "ffn0_bias", "ffn1_weight", "ffn1_bias",
                                          "norm0_weight", "norm0_bias", "norm1_weight", "norm1_bias")]


class DygformerWeights(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("time_w", "time_b", "cooc_w0", "cooc_b0", "cooc_w1", "cooc_b1",
                                           "proj_node_w", "proj_node_b", "proj_edge_w", "proj_edge_b",
                                           "proj_time_w", "proj_time_b", "proj_cooc_w", "proj_cooc_b")]
                + [("layers", EncoderLayerWeights * DYGNN_MAX_LAYERS)]
                + [("output_w", C.c_void_p), ("output_b", C.c_void_p)])


class TgatConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("node_feat_dim", "edge_feat_dim", "time_feat_dim", "num_layers", "num_heads", "num_neighbors")]


class TgatLayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("query_w", "key_w", "value_w", "ln_w", "ln_b", "res_w", "res_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")]


class TgatWeights(C.Structure):
    _fields_ = [("time_w", C.c_void_p), ("time_b", C.c_void_p), ("layers", TgatLayerWeights * DYGNN_MAX_LAYERS)]


class TgatLevels(C.Structure):
    _fields_ = [("ids", C.c_void_p * (DYGNN_MAX_LAYERS + 1)), ("nbr_eid", C.c_void_p * (DYGNN_MAX_LAYERS + 1)),
                ("nbr_dt", C.c_void_p * (DYGNN_MAX_LAYERS + 1))]


class GruWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]


class TgnState(C.Structure):
    _fields_ = [("num_nodes", C.c_int64)] + [(n, C.c_void_p) for n in ("memory", "last_update", "msg", "msg_time", "has_msg")]


class DygformerTaps(C.Structure):
    _fields_ = [("seq_lens", C.c_void_p), ("encoder_input", C.c_void_p), ("layer_out", C.c_void_p * DYGNN_MAX_LAYERS),
                ("phase_cycles", C.c_void_p), ("ev_kernel_start", C.c_void_p), ("ev_kernel_stop", C.c_void_p)]


# name -> (restype, argtypes).  Every symbol include/dygnn.h declares; tests check the export list.
SIGNATURES = {
    "dygnn_last_error": (C.c_char_p, []),
    "dygnn_abi_version": (C.c_int, []),
    "dygnn_mt19937_choice_rows_host": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "dygnn_csr_build_host": (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dygnn_find_neighbors_before": (C.c_int, [C.POINTER(Csr), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                              C.c_void_p]),
    "dygnn_sample_recent": (C.c_int, [C.POINTER(Csr), C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "dygnn_gather_selected": (C.c_int, [C.POINTER(Csr), C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "dygnn_window_lengths": (C.c_int, [C.POINTER(Csr), C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "dygnn_window_fill": (C.c_int, [C.POINTER(Csr), C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dygnn_cooccurrence": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "dygnn_dygformer_packed_bytes": (C.c_size_t, [C.POINTER(DygformerConfig)]),
    "dygnn_dygformer_pack": (C.c_int, [C.POINTER(DygformerConfig), C.POINTER(DygformerWeights), C.c_void_p, C.c_size_t,
                                       C.c_void_p]),
    "dygnn_dygformer_repack": (C.c_int, [C.POINTER(DygformerConfig), C.POINTER(DygformerWeights), C.c_void_p, C.c_size_t,
                                         C.c_int32, C.c_void_p]),
    "dygnn_dygformer_workspace_bytes": (C.c_size_t, [C.POINTER(DygformerConfig), C.c_int64]),
    "dygnn_dygformer_workspace_bytes_for": (C.c_size_t, [C.POINTER(DygformerConfig), C.c_int64, C.c_int32]),
    "dygnn_dygformer_forward": (C.c_int, [C.POINTER(DygformerConfig), C.POINTER(DygformerWeights), C.c_void_p, C.POINTER(Csr),
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(DygformerTaps),
                                          C.c_int32, C.c_void_p]),
    "dygnn_tgat_workspace_bytes": (C.c_size_t, [C.POINTER(TgatConfig), C.c_int64]),
    "dygnn_tgat_forward_levels": (C.c_int, [C.POINTER(TgatConfig), C.POINTER(TgatWeights), C.POINTER(TgatLevels), C.c_void_p, C.c_void_p, C.c_int64,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dygnn_tgat_forward": (C.c_int, [C.POINTER(TgatConfig), C.POINTER(TgatWeights), C.POINTER(Csr), C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dygnn_tgat_forward_roots": (C.c_int, [C.POINTER(TgatConfig), C.POINTER(TgatWeights), C.POINTER(Csr), C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dygnn_tgn_workspace_bytes": (C.c_size_t, [C.POINTER(TgatConfig), C.c_int64, C.c_int64]),
    "dygnn_tgn_forward": (C.c_int, [C.POINTER(TgatConfig), C.POINTER(TgatWeights), C.POINTER(GruWeights), C.POINTER(Csr), C.c_void_p, C.c_void_p,
                                    C.POINTER(TgnState), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dygnn_tgn_forward_step": (C.c_int, [C.POINTER(TgatConfig), C.POINTER(TgatWeights), C.POINTER(GruWeights), C.POINTER(Csr), C.c_void_p, C.c_void_p,
                                         C.POINTER(TgnState), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dygnn_tgn_forward_levels": (C.c_int, [C.POINTER(TgatConfig), C.POINTER(TgatWeights), C.POINTER(GruWeights), C.POINTER(TgatLevels), C.c_void_p, C.c_void_p,
                                           C.POINTER(TgnState), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dygnn_dygformer_train_workspace_bytes": (C.c_size_t, [C.POINTER(DygformerConfig), C.c_int64]),
    "dygnn_dygformer_train_forward": (C.c_int, [C.POINTER(DygformerConfig), C.POINTER(DygformerWeights), C.POINTER(Csr), C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_uint64, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dygnn_dygformer_backward": (C.c_int, [C.POINTER(DygformerConfig), C.POINTER(DygformerWeights), C.POINTER(DygformerWeights), C.c_void_p,
                                           C.c_void_p, C.c_int64, C.c_float, C.c_uint64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "dygnn_merge_layer_sigmoid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dygnn_merge_layer_logits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dygnn_merge_layer_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dygnn_tgat_level_entries": (C.c_int, [C.POINTER(TgatConfig), C.c_int64, C.c_void_p, c_i64p, c_i64p, C.c_void_p]),
    "dygnn_link_metrics_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "dygnn_link_metrics": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.c_void_p]),
}

_lib: Optional[C.CDLL] = None


def lib_path() -> str:
    # DYGNN_LIB_VARIANT=stamps selects the diagnostic build with in-kernel phase stamps (tools/phase_profile.py)
    return _build.lib_path(os.environ.get("DYGNN_LIB_VARIANT", ""))


def load() -> C.CDLL:
    """Load libdygnn_hip.so (building it in-tree first if it is absent and hipcc exists)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        try:
            _build.build(verbose=False, variant=os.environ.get("DYGNN_LIB_VARIANT", ""))
        except Exception as e:  # no hipcc, or compile error: fail loudly, never fall back
            raise DygnnError(f"{path} is missing and could not be built: {e}") from e
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.dygnn_abi_version() != ABI_VERSION:
        raise DygnnError(f"{path}: ABI version {lib.dygnn_abi_version()} != {ABI_VERSION}; rebuild (python -m dyglib_amd._build --force)")
    _lib = lib
    return lib


def load_variant(variant: str) -> C.CDLL:
    """A second, independently loaded build of the library (dyglib_amd/_build.py VARIANTS) next to the default one: the A/B tools time
    several builds of one kernel in ONE process (cdna_hip_programming.md §5.4 rule 24).  Not used by the product path."""
    path = _build.build(verbose=False, variant=variant)
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def check(rc: int, invalid_exc=AssertionError):
    """Map a status code to the exception type the reference raises at the same site
    (AssertionError for invariant violations, e.g. utils/utils.py:157, models/DyGFormer.py:209)."""
    if rc == 0:
        return
    msg = load().dygnn_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise invalid_exc(msg)
    if rc == -3:
        raise NotImplementedError(msg)
    raise DygnnError(f"dygnn error {rc}: {msg}")


def ptr(t) -> Optional[int]:
    """Device/host address of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        return t.data_ptr()
    return t.ctypes.data


def current_stream_ptr() -> Optional[int]:
    return torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else None
