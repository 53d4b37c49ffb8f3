"""ctypes binding of libgridmm_hip.so (C-ABI declared in include/gridmm.h).

The product path FAILS LOUDLY when the library is missing: there is no CPU or
PyTorch fallback behind these entry points.
"""
import ctypes
import os

# ORDER MATTERS: PyTorch-ROCm bundles its own libamdhip64.so.7.  It must be in the process before
# libgridmm_hip.so is dlopen'ed so that the library's DT_NEEDED libamdhip64.so.7 resolves (by SONAME) to the
# SAME HIP runtime that owns torch's streams and allocations; loading ours first binds /opt/rocm's copy and
# every launch on a torch stream then fails with hipErrorNoDevice.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgridmm_hip.so")
# Development build (`make -C gridmm_amd/csrc debug`, -DGRIDMM_DEBUG_HOOKS): the same kernels plus the process-global tuning
# overrides and the whole experiment table of tile configurations.  Only the sweep tools ask for it (load(debug=True) or
# GRIDMM_LIB_DEBUG=1 before the first load); the product path and the tests run on the shipping library, which has neither.
DEBUG_LIB_PATH = os.path.join(_HERE, "libgridmm_hip_dbg.so")
ABI_VERSION = 29

_vp, _i, _f, _i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64

# name -> argtypes, exactly the prototypes of include/gridmm.h
SIGNATURES = {
    "gridmm_abi_version": [],
    "gridmm_hbm_read_probe": [_vp, ctypes.c_size_t, _vp, _vp],
    "gridmm_grid_project": [_vp, _i, _vp, _vp, _vp, _i] + [_vp] * 9 + [_i, _i, _i, _i, _f, _i, _f, _vp],
    "gridmm_grid_bin": [_vp] * 10 + [_i, _i, _i, _vp],
    "gridmm_grid_bin_sliced": [_vp] * 11 + [_i, _i, _i, _i, _vp],
    "gridmm_grid_sort_ids": [_vp] * 4 + [_i, _i, _vp],
    "gridmm_grid_cell_count_max": [_vp, _vp, _i, _vp],
    "gridmm_text_fragments": [_vp, _vp, _i, _i, _i, _vp],
    "gridmm_grid_aggregate_workspace": [_i, _i, _i],
    "gridmm_grid_aggregate": [_vp] * 8 + [_i, _i, _i, _i, _i, _vp],
    "gridmm_grid_aggregate_train": [_vp] * 9 + [_i, _i, _i, _i, _i, _vp],
    "gridmm_grid_aggregate_incremental_scratch": [_i, _i],
    "gridmm_grid_aggregate_incremental": [_vp] * 6 + [_i] + [_vp] * 7 + [_i, _i, _i, _i, _i, _i, _vp],
    "gridmm_cells_compact": [_vp] * 7 + [_i, _i, _i, _vp],
    "gridmm_split_weight": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "gridmm_linear": [_vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "gridmm_layernorm": [_vp, _i, _vp, _i, _vp, _vp, _f, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "gridmm_split_rows": [_vp, _i, _vp, _vp, _i, _i, _i, _vp],
    "gridmm_linear_planes": [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "gridmm_linear_planes_cfg": [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "gridmm_attention": [_vp, _i64, _i, _vp, _i64, _i, _vp, _i64, _i, _vp, _i, _vp, _i64, _i,
                         _vp, _vp, _i64, _i, _i, _i, _i, _i, _f, _vp],
    "gridmm_transpose_v": [_vp, _vp, _i64, _i, _vp, _vp, _i, _i, _i, _i, _vp],
    "gridmm_attention_planes": [_vp, _vp, _i64, _i, _vp, _vp, _i64, _i, _vp, _vp, _i, _vp, _i, _vp, _i64, _i,
                                _vp, _vp, _i64, _i, _i, _i, _i, _i, _f, _vp],
    "gridmm_attention_rows": [_vp, _vp, _i64, _i, _vp, _vp, _i64, _i, _vp, _vp, _i64, _i, _vp, _i, _vp, _i64, _i,
                              _vp, _vp, _i64, _i, _i, _i, _i, _i, _f, _vp],
    "gridmm_attention_rows_cfg": [_vp, _vp, _i64, _i, _vp, _vp, _i64, _i, _vp, _vp, _i64, _i, _vp, _i, _vp, _i64, _i,
                                  _vp, _vp, _i64, _i, _i, _i, _i, _i, _f, _i, _vp],
    "gridmm_xattn_layer_workspace": [_i, _i, _i, _i],
    "gridmm_xattn_layer_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp, _vp, _i64, _i, _i, _i, _vp, _i, _vp, _i,
                               _vp, _vp, _vp, _i, _i64, _vp, ctypes.c_size_t, _i, _i, _i, _i, _vp],
    "gridmm_attention_rows_seg": [_vp, _vp, _i64, _i, _vp, _vp, _i64, _i, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _i64, _i,
                                  _vp, _i, _vp, _i64, _i, _vp, _vp, _i64, _i, _i, _i, _i, _i, _f, _vp],
    "gridmm_linear_planes_tn": [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp],
    "gridmm_linear_planes_tn_db": [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "gridmm_linear_planes_tn_grouped": [_vp, _i, _vp],
    "gridmm_linear_planes_tn_splits": [_i, _i, _i],
    "gridmm_split_rows_pad": [_vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp],
    "gridmm_linear_planes_map": [_vp, _vp, _i, _i, _i64, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "gridmm_linear_planes_grouped": [_vp, _i, _vp],
    "gridmm_layernorm_map": [_vp, _i, _vp, _i, _vp, _vp, _f, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i64, _i, _i, _vp],
    "gridmm_split_rows_map": [_vp, _i, _vp, _vp, _i, _i, _i64, _i, _i, _vp],
    "gridmm_cells_embed": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp],
    "gridmm_node_embed": [_vp, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp],
    "gridmm_nav_heads": [_vp, _i] + [_vp] * 17 + [_i, _i, _i, _i, _vp],
    "gridmm_tokens_to_slab": [_vp, _i, _i, _vp, _i64, _i, _i, _vp],
    "gridmm_ln_dot": [_vp, _i, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _vp],
    "gridmm_fuse_logits": [_vp] * 13 + [_i, _i, _i, _vp],
    "gridmm_copy_rows": [_vp, _i64, _i, _vp, _i64, _i, _i, _i, _i, _vp],
    # training (backward)
    "gridmm_transpose_split": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "gridmm_layernorm_bwd": [_vp, _i, _vp, _i, _vp, _f, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _vp],
    "gridmm_layernorm_bwd_planes": [_vp, _i, _vp, _i, _vp, _f, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "gridmm_activation": [_vp, _vp, _vp, _i64, _i, _vp],
    "gridmm_activation_planes": [_vp, _vp, _vp, _vp, _vp, _i64, _i, _vp],
    "gridmm_attention_train_planes": [_vp, _i64, _i, _vp, _i64, _i, _vp, _i64, _i, _vp, _i, _vp, _i64, _i, _vp, _vp, _i64, _i,
                                      _vp, _i, _i, _i, _i, _i, _f, _f, ctypes.c_uint64, _vp, _vp],
    "gridmm_layernorm_dropout_planes": [_vp, _vp, _i, _vp, _vp, _f, _vp, _vp, _vp, _f, ctypes.c_uint64, _vp, _i, _i, _vp],
    "gridmm_attention_train": [_vp, _i64, _i, _vp, _i64, _i, _vp, _i64, _i, _vp, _i, _vp, _i64, _i, _vp, _i,
                               _i, _i, _i, _i, _f, _f, ctypes.c_uint64, _vp, _vp],
    "gridmm_attention_bwd": [_vp, _i64, _i, _vp, _i64, _i, _vp, _i64, _i, _vp, _i, _vp, _i64, _i, _vp, _i64, _i,
                             _vp, _vp, _vp, _i64, _i, _vp, _i64, _i, _vp, _i64, _i, _i, _i, _i, _i, _i, _f, _f,
                             ctypes.c_uint64, _vp, _vp],
    "gridmm_attention_rows_train": [_vp, _vp, _i64, _i, _vp, _vp, _i64, _i, _vp, _vp, _i64, _i, _vp, _i, _vp, _i64, _i, _vp, _vp,
                                    _i64, _i, _vp, _i, _vp, _i64, _i, _i, _i, _i, _f, _f, ctypes.c_uint64, _vp, _vp],
    "gridmm_linear_planes_shift": [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "gridmm_attention_rows_bwd": [_vp, _vp, _i64, _i, _vp, _vp, _i64, _i, _vp, _vp, _i64, _i, _vp, _i, _vp, _i64, _i, _vp, _i64, _i,
                                  _vp, _vp, _i64, _vp, ctypes.c_size_t, _vp, _i64, _i, _vp, _i64, _i, _vp, _i64, _i, _i, _i, _i, _i, _i, _f, _f,
                                  ctypes.c_uint64, _vp, _vp],
    "gridmm_attention_rows_bwd_planes": [_vp, _vp, _i64, _i, _vp, _vp, _i64, _i, _vp, _vp, _i64, _i, _vp, _i, _vp, _i64, _i, _vp, _i64, _i,
                                  _vp, _vp, _i64, _vp, ctypes.c_size_t, _vp, _i64, _i, _vp, _i64, _i, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f,
                                  ctypes.c_uint64, _vp, _vp],
    "gridmm_grid_aggregate_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "gridmm_grid_aggregate_bwd_routed": [_vp] * 10 + [_i, _i, _i, _i, _vp],
    "gridmm_fuse_logits_bwd": [_vp] * 16 + [_i, _i, _i, _vp],
    "gridmm_cells_compact_bwd": [_vp, _i64, _vp, _vp, _i, _i, _vp],
    "gridmm_dropout": [_vp, _vp, _i64, _f, ctypes.c_uint64, _vp, _vp],
    "gridmm_layernorm_dropout": [_vp, _vp, _i, _vp, _vp, _f, _vp, _f, ctypes.c_uint64, _vp, _i, _i, _vp],
    "gridmm_layernorm_dropout_bwd": [_vp, _vp, _i, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _f, ctypes.c_uint64, _vp, _i, _i, _vp],
    "gridmm_layernorm_dropout_bwd_planes": [_vp, _vp, _i, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, ctypes.c_uint64, _vp, _i,
                                            _i, _vp],
    "gridmm_grad_sumsq": [_vp, _i64, _i, _vp, _vp],
    "gridmm_adamw_step": [_vp, _vp, _vp, _vp, _i64, _i, _f, _f, _f, _f, _f, _f, _i, _vp, _f, _vp, _vp],
    "gridmm_linear_planes_splitk": [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp],
    "gridmm_multi_grad_sumsq": [_vp, _vp, _i, _i, _vp, _vp, _vp],
    "gridmm_multi_adamw_step": [_vp, _vp, _i, _i, _f, _f, _i, _vp, _f, _vp, _vp],
    "gridmm_multi_grad_accumulate": [_vp, _vp, _i, _i, _vp],
    "gridmm_xattn_layer_train_saved_bytes": [_i, _i, _i, _i],
    "gridmm_xattn_layer_train_workspace": [_i, _i, _i, _i],
    "gridmm_xattn_layer_train_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, ctypes.c_size_t, _vp,
                                     ctypes.c_size_t, _i, _i, _i, _i, _vp],
    "gridmm_xattn_layer_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i, _i, _i, _vp, _i, _vp, _i, _vp, ctypes.c_size_t, _vp, _vp, _vp, _i64, _i,
                               _vp, _vp, ctypes.c_size_t, _i, _i, _i, _i, _vp],
    "gridmm_preln_layer_saved_bytes": [_i, _i, _i, _i],
    "gridmm_preln_layer_workspace": [_i, _i, _i, _i],
    "gridmm_preln_layer_train_fwd": [_vp, _vp, _vp, _i, _vp, _vp, ctypes.c_size_t, _vp, ctypes.c_size_t, _i, _i, _i, _vp],
    "gridmm_preln_layer_bwd": [_vp, _vp, _vp, _i, _vp, ctypes.c_size_t, _vp, _vp, _vp, _vp, ctypes.c_size_t, _i, _i, _i, _vp],
    "gridmm_dropout_add": [_vp, _vp, _vp, _vp, _vp, _i64, _f, ctypes.c_uint64, _vp, _vp],
    "gridmm_rowdot": [_vp, _i, _vp, _vp, _vp, _i, _i, _vp],
    "gridmm_rowdot_bwd_workspace": [_i, _i],
    "gridmm_rowdot_bwd": [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp],
    "gridmm_linear_skinny": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "gridmm_linear_skinny_bwd_workspace": [_i, _i, _i],
    "gridmm_linear_skinny_bwd": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp],
    # host-side helpers of the agent loop (no device work)
    "gridmm_route_lengths": [_vp, _i, _i, _vp, _vp, _vp, _i, _vp],
    "gridmm_collate_nav_plan": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "gridmm_collate_nav_fill": [_vp] * 15 + [_i] * 8 + [_vp] * 10,
}

# development build only (GRIDMM_DEBUG_HOOKS section of include/gridmm.h)
DEBUG_SIGNATURES = {
    "gridmm_debug_gemm_cfg_override": [_i, _i, _i, _i],
    "gridmm_debug_attention_cfg_override": [_i, _i],
    "gridmm_debug_gemm_shapes": [_vp, _i, _i],
    "gridmm_debug_linear_planes_map_cfg": [_vp, _vp, _i, _i, _i64, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i,
                                           _i, _i, _i, _vp],
}
W_ROWMAJOR, W_TILED = 0, 1

_lib = None
_lib_is_debug = False


class GridmmLibraryError(RuntimeError):
    pass


def load(debug=None):
    """dlopen the HIP library (no GPU needed for loading) and bind every prototype.  debug=True (or GRIDMM_LIB_DEBUG=1 in the
    environment at the first call) binds the development build instead; one library per process."""
    global _lib, _lib_is_debug
    if debug is None:
        debug = _lib_is_debug if _lib is not None else os.environ.get("GRIDMM_LIB_DEBUG", "0") not in ("", "0")
    if _lib is not None:
        if debug and not _lib_is_debug:
            raise GridmmLibraryError("the shipping library is already loaded in this process; set GRIDMM_LIB_DEBUG=1 (or call "
                                     "load(debug=True)) before the first use to get the development build")
        return _lib
    path = DEBUG_LIB_PATH if debug else LIB_PATH
    if not os.path.exists(path):
        raise GridmmLibraryError(
            "%s not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C gridmm_amd/csrc%s`). "
            "There is no fallback path." % (path, " debug" if debug else ""))
    lib = ctypes.CDLL(path)
    sigs = dict(SIGNATURES)
    if debug:
        sigs.update(DEBUG_SIGNATURES)
    for name, argtypes in sigs.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    lib.gridmm_xattn_layer_workspace.restype = ctypes.c_size_t
    lib.gridmm_grid_aggregate_bwd_workspace.argtypes = [_i, _i, _i]
    lib.gridmm_grid_aggregate_bwd_workspace.restype = ctypes.c_size_t
    lib.gridmm_attention_rows_bwd_workspace.argtypes = [_i, _i, _i]
    lib.gridmm_attention_rows_bwd_workspace.restype = ctypes.c_size_t
    lib.gridmm_grid_aggregate_workspace.restype = ctypes.c_size_t
    lib.gridmm_grid_aggregate_incremental_scratch.restype = ctypes.c_size_t
    lib.gridmm_xattn_layer_train_saved_bytes.restype = ctypes.c_size_t
    lib.gridmm_xattn_layer_train_workspace.restype = ctypes.c_size_t
    lib.gridmm_preln_layer_saved_bytes.restype = ctypes.c_size_t
    lib.gridmm_preln_layer_workspace.restype = ctypes.c_size_t
    lib.gridmm_linear_skinny_bwd_workspace.restype = ctypes.c_size_t
    lib.gridmm_rowdot_bwd_workspace.restype = ctypes.c_size_t
    lib.gridmm_nav_heads_workspace.argtypes = [_i, _i, _i]
    lib.gridmm_nav_heads_workspace.restype = ctypes.c_size_t
    v = lib.gridmm_abi_version()
    if v != ABI_VERSION:
        raise GridmmLibraryError("%s ABI %d != expected %d (stale build?)" % (os.path.basename(path), v, ABI_VERSION))
    _lib, _lib_is_debug = lib, bool(debug)
    return lib


def check(status, what):
    if status != 0:
        detail = " (hipError_t %d)" % (-1000 - status) if status <= -1000 else ""
        raise GridmmLibraryError("%s failed with status %d%s" % (what, status, detail))
