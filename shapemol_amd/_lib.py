"""ctypes binding of libshapemol_hip.so (C ABI: include/shapemol_hip.h).

The library is built in-tree by ``shapemol_amd/csrc/build.sh`` (hipcc, gfx950).  There is no
CPU fallback: if the library cannot be loaded, every entry point of the package raises.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# SHAPEMOL_LIB selects a diagnostic build (tools/ only), e.g. "stamps" or "abl1"
_variant = os.environ.get("SHAPEMOL_LIB") or ("stamps" if os.environ.get("SHAPEMOL_STAMPS") == "1" else "")
LIB_PATH = os.path.join(_HERE, f"libshapemol_hip_{_variant}.so" if _variant else "libshapemol_hip.so")
ABI_VERSION = 5

EXPORTS = (
    "shapemol_abi_version", "shapemol_last_error", "shapemol_weight_count", "shapemol_create",
    "shapemol_destroy", "shapemol_reserve", "shapemol_score", "shapemol_sample",
    "shapemol_log_sample_categorical", "shapemol_set_option", "shapemol_debug_read",
    "shapemol_profile_begin", "shapemol_profile_end", "shapemol_status", "shapemol_status_stream", "shapemol_set_guidance", "shapemol_guide_points",
    "shapemol_pointcloud_guidance", "shapemol_set_knn_pins", "shapemol_debug_split_exact",
    "shapemol_mlp_backward_workspace", "shapemol_mlp_forward", "shapemol_mlp_backward",
    "shapemol_seg_attention_forward", "shapemol_seg_attention_backward",
    "shapemol_vn_backward_workspace", "shapemol_vn_forward", "shapemol_vn_backward",
    "shapemol_edge_mlp_backward_workspace", "shapemol_edge_mlp_forward", "shapemol_edge_mlp_backward",
    "shapemol_set_bn_running",
    "shapemol_se_weight_count", "shapemol_se_create", "shapemol_se_destroy", "shapemol_se_encode",
)


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "hidden_dim", "n_heads", "num_layers", "knn", "num_r_gaussian", "shape_dim",
        "shape_latent_dim", "time_emb_dim", "num_classes", "num_timesteps")]


class Traj(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "pos_traj", "v_traj", "v0_traj", "vt_traj", "pos_cond_traj", "v_cond_traj")]


class ShapeMolLibraryError(RuntimeError):
    pass


def build(verbose=False):
    """Compile the HIP library in-tree (needs hipcc; cross-compiles for gfx950 without a GPU)."""
    script = os.path.join(_HERE, "csrc", "build.sh")
    r = subprocess.run(["bash", script], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(LIB_PATH):
        raise ShapeMolLibraryError("building libshapemol_hip.so failed:\n" + r.stdout + r.stderr)
    if verbose:
        print(r.stdout + r.stderr)
    return LIB_PATH


_lib = None


def load():
    """Load (once) and type the library.  Raises ShapeMolLibraryError if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ShapeMolLibraryError(
            f"{LIB_PATH} not found: build it with `bash shapemol_amd/csrc/build.sh` "
            "(or __graft_entry__.build()).  shapemol_amd has no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise ShapeMolLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    vp, i64, i32, u64 = C.c_void_p, C.c_int64, C.c_int32, C.c_uint64
    lib.shapemol_abi_version.restype = C.c_int
    lib.shapemol_last_error.restype = C.c_char_p
    lib.shapemol_weight_count.restype = C.c_size_t
    lib.shapemol_weight_count.argtypes = [C.POINTER(Config)]
    lib.shapemol_create.argtypes = [C.POINTER(Config), vp, C.c_size_t, C.c_int, C.POINTER(vp)]
    lib.shapemol_destroy.argtypes = [vp]
    lib.shapemol_destroy.restype = None
    lib.shapemol_reserve.argtypes = [vp, i64, i64]
    lib.shapemol_score.argtypes = [vp, vp, vp, vp, i64, i64, vp, vp, vp, vp, vp, vp]
    lib.shapemol_sample.argtypes = [vp, vp, vp, vp, i64, i64, vp, i32, vp, vp, u64, C.POINTER(Traj), vp, vp, i32, vp]
    lib.shapemol_log_sample_categorical.argtypes = [vp, vp, vp, i64, i32, u64, vp, vp]
    lib.shapemol_set_option.argtypes = [vp, C.c_char_p, i64]
    lib.shapemol_debug_read.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
    lib.shapemol_debug_read.restype = i64
    lib.shapemol_status.argtypes = [vp, vp]
    lib.shapemol_status_stream.argtypes = [vp, vp, vp]
    lib.shapemol_set_bn_running.argtypes = [vp, vp, vp, i64]
    lib.shapemol_set_guidance.argtypes = [vp, vp, i64, C.c_double, i32, vp]
    lib.shapemol_guide_points.argtypes = [vp, vp, i64, vp, u64, vp]
    lib.shapemol_set_knn_pins.argtypes = [vp, vp, i32, vp, vp, i64, i32]
    lib.shapemol_debug_split_exact.argtypes = [C.c_float, vp]
    lib.shapemol_debug_split_exact.restype = None
    lib.shapemol_pointcloud_guidance.argtypes = [vp, i64, C.c_double, C.c_double, vp, i64, vp, u64, vp]
    lib.shapemol_mlp_backward_workspace.restype = C.c_size_t
    lib.shapemol_mlp_backward_workspace.argtypes = [i64, i32, i32, i32]
    lib.shapemol_mlp_forward.argtypes = [vp, i64, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.shapemol_mlp_backward.argtypes = [vp, vp, i64, i32, i32, i32] + [vp] * 15 + [C.c_size_t, vp]
    lib.shapemol_edge_mlp_backward_workspace.restype = C.c_size_t
    lib.shapemol_edge_mlp_backward_workspace.argtypes = [i64, i64, i32, i32, i32, i32, i32]
    lib.shapemol_edge_mlp_forward.argtypes = [vp] * 5 + [i64, i64] + [i32] * 5 + [vp] * 13
    lib.shapemol_edge_mlp_backward.argtypes = [vp] * 7 + [i64, i64] + [i32] * 5 + [vp] * 17 + [C.c_size_t, vp]
    lib.shapemol_vn_backward_workspace.restype = C.c_size_t
    lib.shapemol_vn_backward_workspace.argtypes = [i64, i32, i32, i32]
    lib.shapemol_vn_forward.argtypes = [vp] * 4 + [i64, i32, i32, i32] + [vp] * 6 + [i32] + [vp] * 6
    lib.shapemol_vn_backward.argtypes = [vp] * 4 + [i64, i32, i32, i32] + [vp] * 7 + [i32] + [vp] * 8 + [C.c_size_t, vp]
    lib.shapemol_seg_attention_forward.argtypes = [vp, vp, vp, vp, i64, i32, i32, i32, vp, vp]
    lib.shapemol_seg_attention_backward.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, i32, vp, vp, vp, vp]
    lib.shapemol_se_weight_count.restype = C.c_size_t
    lib.shapemol_se_weight_count.argtypes = [i32, i32, i32]
    lib.shapemol_se_create.argtypes = [i32, i32, i32, i32, vp, C.c_size_t, C.c_int, C.POINTER(vp)]
    lib.shapemol_se_destroy.argtypes = [vp]
    lib.shapemol_se_destroy.restype = None
    lib.shapemol_se_encode.argtypes = [vp, vp, i64, i64, vp, vp]
    lib.shapemol_profile_begin.argtypes = [vp]
    lib.shapemol_profile_end.argtypes = [vp, vp, vp, vp, C.c_int]
    if lib.shapemol_abi_version() != ABI_VERSION:
        raise ShapeMolLibraryError("libshapemol_hip.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().shapemol_last_error()
        raise ShapeMolLibraryError(f"{what} failed: {msg.decode() if msg else rc}")
