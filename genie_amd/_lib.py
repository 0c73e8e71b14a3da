"""ctypes binding of libgenie_hip.so (the C ABI declared in include/genie_hip.h).

The product path has NO fallback: if the shared library is missing or cannot be loaded this module
raises, and every op that needs it raises with it. Build it with `python __graft_entry__.py build`
(or `genie_amd._lib.build()`), which runs `hipcc --offload-arch=gfx950` on `genie_amd/csrc/genie_hip.hip`.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(_HERE)
SRC = os.path.join(_HERE, "csrc", "genie_hip.hip")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.environ.get("GENIE_LIB_PATH", os.path.join(LIB_DIR, "libgenie_hip.so"))  # override: tuning builds (tools/tune.py) only
INCLUDE = os.path.join(REPO, "include")

# every symbol include/genie_hip.h declares: (name, restype, argtypes)
_c = ctypes
_P = _c.c_void_p
SYMBOLS = [
    ("genie_version", _c.c_int, []),
    ("genie_last_error", _c.c_char_p, []),
    ("genie_ctx_create", _c.c_int, [_c.POINTER(_P), _c.c_int, _c.c_int, _c.c_int, _P, _P, _P, _P, _P, _c.c_float]),
    ("genie_ctx_create_subgraph", _c.c_int, [_c.POINTER(_P), _c.c_int, _c.c_int, _c.c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _c.c_float]),
    ("genie_ctx_destroy", _c.c_int, [_P]),
    ("genie_set_scale_t", _c.c_int, [_P, _c.c_float]),
    ("genie_set_edge_features", _c.c_int, [_P, _P, _P, _P]),
    ("genie_set_absolute_pos", _c.c_int, [_P, _P, _P, _P]),
    ("genie_nbr_mean", _c.c_int, [_P, _P, _P, _P, _P, _c.c_int, _P]),
    ("genie_nbr_mean_bwd", _c.c_int, [_P, _P, _P, _P, _P, _c.c_int, _P]),
    ("genie_prelu_bwd", _c.c_int, [_P, _P, _P, _c.c_int64, _P, _P, _P, _P]),
    ("genie_linear_bwd_scratch_floats", _c.c_int64, [_c.c_int]),
    ("genie_linear_bwd_wb", _c.c_int, [_P, _P, _c.c_int64, _c.c_int, _c.c_int, _P, _P, _P, _P]),
    ("genie_set_slot", _c.c_int, [_P, _c.c_int]),
    ("genie_set_station_order", _c.c_int, [_P, _P]),
    ("genie_set_phase_types", _c.c_int, [_P, _c.c_int]),
    ("genie_set_sign_input", _c.c_int, [_P, _c.c_int]),
    ("genie_set_stage2_workmap", _c.c_int, [_P, _c.c_int]),
    ("genie_set_subgraph_stations", _c.c_int, [_P, _P, _P]),
    ("genie_set_tail_precision", _c.c_int, [_P, _c.c_int]),
    ("genie_set_stage_precision", _c.c_int, [_P, _c.c_int]),
    ("genie_stage_precision", _c.c_int, [_P, _c.POINTER(_c.c_int), _c.POINTER(_c.c_int), _c.POINTER(_c.c_float), _c.POINTER(_c.c_float), _P]),
    ("genie_input_range", _c.c_int, [_P, _c.POINTER(_c.c_float), _c.POINTER(_c.c_float), _c.c_int]),
    ("genie_index_flags", _c.c_int, [_P, _c.POINTER(_c.c_uint), _c.c_int]),
    ("genie_index_check", _c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_uint, _P]),
    ("genie_set_static_edge_attr", _c.c_int, [_P, _P, _P]),
    ("genie_tail_batched", _c.c_int, [_P, _c.c_int, _c.c_int, _P, _P, _P, _c.c_int, _c.c_int, _P, _c.c_int, _P, _P, _P, _P, _P]),
    ("genie_readout_grid", _c.c_int, [_P, _P, _P, _c.c_int, _P, _P]),
    ("genie_readout_query", _c.c_int, [_P, _P, _P, _P, _P, _c.c_int, _c.c_int, _P, _c.c_int, _P, _P, _P]),
    ("genie_readout_grid_latent", _c.c_int, [_P, _P, _P, _c.c_int, _P, _P, _P]),
    ("genie_readout_query_latent", _c.c_int, [_P, _P, _P, _P, _P, _c.c_int, _c.c_int, _P, _c.c_int, _P, _P, _P, _P]),
    ("genie_weights_count", _c.c_int, []),
    ("genie_weights_name", _c.c_char_p, [_c.c_int]),
    ("genie_weights_numel", _c.c_int64, [_c.c_int]),
    ("genie_weights_offset", _c.c_int64, [_c.c_int]),
    ("genie_weights_blob_floats", _c.c_int64, []),
    ("genie_weights_set", _c.c_int, [_P, _c.c_char_p, _P, _c.c_int64, _P]),
    ("genie_weights_set_blob", _c.c_int, [_P, _P, _c.c_int64, _P]),
    ("genie_weights_commit", _c.c_int, [_P, _P]),
    ("genie_workspace_bytes", _c.c_size_t, [_P]),
    ("genie_da_stage1", _c.c_int, [_P, _P, _P, _P, _P]),
    ("genie_da_stage1_range", _c.c_int, [_P, _P, _P, _c.c_int, _c.c_int, _c.c_int, _P, _P]),
    ("genie_da_stage2_partials_range", _c.c_int, [_P, _P, _P, _P, _c.c_int, _c.c_int, _P, _P]),
    ("genie_da_stage1_debug", _c.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    ("genie_ws_v_ptr", _P, [_P, _P]),
    ("genie_ws_v_pitch", _c.c_int, [_P]),
    ("genie_da_stage2_bipartite", _c.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    ("genie_da_stage2_partials", _c.c_int, [_P, _P, _P, _P, _P, _P]),
    ("genie_bipartite_readout", _c.c_int, [_P, _P, _P, _P]),
    ("genie_spatial_agg_fwd", _c.c_int, [_P, _c.c_int, _P, _P, _P, _P, _P]),
    ("genie_spatial_agg3_fwd", _c.c_int, [_P, _P, _P, _P, _P, _P]),
    ("genie_path_fwd", _c.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    ("genie_where_am_i", _c.c_int, [_P, _c.c_int, _P]),
    ("genie_set_tail_grid", _c.c_int, [_P, _c.c_int, _c.c_int]),
    ("genie_train_save_floats", _c.c_size_t, [_P]),
    ("genie_train_scratch_floats", _c.c_size_t, [_P]),
    ("genie_da_train_fwd", _c.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P]),
    ("genie_da_train_bwd", _c.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P]),
    ("genie_tail_train_save_floats", _c.c_size_t, [_P]),
    ("genie_tail_train_scratch_floats", _c.c_size_t, [_P, _c.c_int]),
    ("genie_train_grad_floats", _c.c_size_t, []),
    ("genie_tail_train_fwd", _c.c_int, [_P, _P, _P, _P, _c.c_int, _c.c_int, _P, _c.c_int, _P, _P, _P, _P, _P, _P]),
    ("genie_tail_train_bwd", _c.c_int, [_P, _P, _P, _P, _P, _P, _c.c_int, _c.c_int, _P, _c.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    ("genie_train_bwd", _c.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _c.c_int, _c.c_int, _P, _c.c_int, _P, _P, _P, _P, _P, _P, _P,
                                   _P, _P, _P, _P]),
    ("genie_assoc_workspace_bytes", _c.c_size_t, [_P]),
    ("genie_assoc_fwd", _c.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    ("genie_assoc_train_save_floats", _c.c_size_t, [_P]),
    ("genie_assoc_train_scratch_floats", _c.c_size_t, [_P]),
    ("genie_assoc_train_fwd", _c.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    ("genie_assoc_train_bwd", _c.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    ("genie_knn", _c.c_int, [_P, _c.c_int, _P, _c.c_int, _c.c_int, _c.c_int, _P, _P]),
    ("genie_product_check", _c.c_int, [_P, _c.c_int64, _P, _c.c_int64, _c.c_int, _c.c_int, _P, _P]),
    ("genie_lslc_fwd", _c.c_int, [_P, _c.c_int, _P, _P, _c.c_int64, _c.c_int, _c.c_float, _c.c_float, _P, _c.c_float, _P, _c.c_int, _c.c_int,
                                  _P, _P, _P, _c.c_int, _P, _P]),
    ("genie_lslc_bwd_part_floats", _c.c_size_t, [_c.c_int]),
    ("genie_lslc_bwd", _c.c_int, [_P, _c.c_int, _P, _P, _c.c_int64, _c.c_int, _c.c_float, _c.c_float, _P, _c.c_float, _P, _c.c_int, _c.c_int,
                                  _P, _P, _P, _c.c_int, _P, _P, _P, _P, _P, _P]),
    ("genie_seg_rows", _c.c_int, [_P, _P, _P, _c.c_int64, _P, _P]),
    ("genie_arrivals_fwd", _c.c_int, [_P, _c.c_int, _P, _P, _P, _c.c_int, _P, _P, _P, _P, _c.c_int, _P, _P, _P, _P, _c.c_int, _c.c_float,
                                      _P, _P, _P, _P]),
    ("genie_arrivals_train_save_floats", _c.c_int64, [_c.c_int, _c.c_int]),
    ("genie_arrivals_train_fwd", _c.c_int, [_P, _c.c_int, _P, _P, _P, _c.c_int, _P, _P, _P, _P, _c.c_int, _P, _P, _P, _P, _c.c_int,
                                            _c.c_float, _P, _P, _P, _P, _P]),
    ("genie_arrivals_bwd_scratch_floats", _c.c_int64, [_c.c_int, _c.c_int, _c.c_int]),
    ("genie_arrivals_bwd", _c.c_int, [_P, _c.c_int, _P, _P, _P, _c.c_int, _P, _P, _P, _P, _c.c_int, _P, _P, _P, _P, _c.c_int, _c.c_float,
                                      _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    ("genie_subgraph_csr_count", _c.c_int, [_P, _P, _c.c_int64, _P, _P, _P, _P, _P, _P, _P, _P]),
    ("genie_subgraph_csr_fill", _c.c_int, [_P, _P, _c.c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    ("genie_row_select_count", _c.c_int, [_P, _c.c_int, _c.c_int64, _c.c_float, _c.c_int, _P, _P]),
    ("genie_row_select_fill", _c.c_int, [_P, _c.c_int, _c.c_int64, _c.c_float, _c.c_int, _P, _P, _P, _P, _P]),
    ("genie_ws_export", _c.c_int, [_P, _c.c_int, _P, _P, _P]),
    ("genie_embed_ntime", _c.c_int, [_c.c_double, _c.c_double, _c.c_double, _c.c_double]),
    ("genie_embed_window", _c.c_int, [_P, _P, _P, _P, _c.c_int, _c.c_double, _c.c_double, _c.c_double, _c.c_double, _P, _P,
                                      _P, _P, _P]),
    ("genie_embed_window_split", _c.c_int, [_P, _P, _P, _P, _c.c_int, _c.c_double, _c.c_double, _c.c_double, _c.c_double, _P, _P,
                                            _P, _P, _P, _P]),
]


class GenieHipError(RuntimeError):
    pass


def build(verbose=False, extra_flags=(), out_path=None):
    """Compile the HIP extension for gfx950 in-tree (cross-compiles without a GPU)."""
    os.makedirs(LIB_DIR, exist_ok=True)
    if out_path:
        os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    # -fno-honor-nans: no NaN-canonicalisation v_max before every fmaxf/fminf (no reassociation is enabled);
    # -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs, not AGPRs + v_accvgpr_read copies
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           "-fno-honor-nans", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form",
           "-I", INCLUDE, SRC, "-o", out_path or LIB_PATH] + list(extra_flags)
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise GenieHipError("hipcc failed:\n" + r.stdout)
    return out_path or LIB_PATH


_lib = None


def load():
    """Load libgenie_hip.so and bind every declared symbol. Raises GenieHipError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GenieHipError(
            "libgenie_hip.so not found at %s — the HIP extension is not built. Run "
            "`python __graft_entry__.py build` (needs hipcc); there is no CPU or PyTorch fallback." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise GenieHipError("cannot load %s: %s" % (LIB_PATH, e))
    for name, res, args in SYMBOLS:
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise GenieHipError("libgenie_hip.so does not export %s (stale build?)" % name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().genie_last_error()
        raise GenieHipError("%s failed (%d): %s" % (what or "libgenie_hip call", rc, (msg or b"").decode()))
