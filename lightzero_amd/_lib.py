"""ctypes binding of liblz_mi355.so (include/lz_mi355.h).  There is NO fallback: if the HIP
library is missing or no gfx950 device is visible, compute entry points raise."""
import ctypes
import os
import sys

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "liblz_mi355.so")


class LzError(RuntimeError):
    pass


class ModelCfg(ctypes.Structure):
    _fields_ = [("model_type", ctypes.c_int), ("obs_c", ctypes.c_int), ("obs_h", ctypes.c_int), ("obs_w", ctypes.c_int),
                ("action_space_size", ctypes.c_int), ("num_channels", ctypes.c_int), ("lstm_hidden_size", ctypes.c_int),
                ("head_channels", ctypes.c_int), ("head_hidden", ctypes.c_int), ("support_size", ctypes.c_int),
                ("support_min", ctypes.c_float), ("bn_eps", ctypes.c_float), ("downsample", ctypes.c_int),
                ("activation", ctypes.c_int), ("res_connection_in_dynamics", ctypes.c_int), ("action_encoding", ctypes.c_int),
                ("num_of_sampled_actions", ctypes.c_int), ("sigma_type", ctypes.c_int), ("bound_type", ctypes.c_int),
                ("ln_eps", ctypes.c_float)]


_lib = None
c_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
c_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
P = ctypes.c_void_p


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LzError("%s not found: build it with `python -m lightzero_amd.build` (hipcc, gfx950). "
                      "lightzero_amd has no CPU/PyTorch fallback." % LIB_PATH)
    if "torch" not in sys.modules and not os.environ.get("LZ_NO_TORCH_PRELOAD"):
        # PyTorch-ROCm bundles its own libamdhip64; when both live in one process the HIP runtime must be
        # loaded once.  Importing torch first makes this library bind to the runtime torch already mapped.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = ctypes.CDLL(LIB_PATH)
    L.lz_last_error.restype = ctypes.c_char_p
    L.lz_engine_stream.restype = P
    sig = {
        "lz_engine_create": [ctypes.c_int, ctypes.POINTER(P)],
        "lz_engine_destroy": [P],
        "lz_engine_synchronize": [P],
        "lz_engine_stream": [P],
        "lz_roots_create": [P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_i32p, c_i32p, ctypes.POINTER(P)],
        "lz_roots_destroy": [P],
        "lz_roots_reset": [P, c_i32p, c_i32p],
        "lz_roots_reset_keep_inference": [P, c_i32p, c_i32p],
        "lz_roots_minmax_reset": [P, ctypes.c_float],
        "lz_roots_set_tiebreak": [P, ctypes.c_int, ctypes.c_uint64],
        "lz_roots_prepare": [P, ctypes.c_float, P, c_f32p, c_f32p, c_i32p],
        "lz_roots_prepare_device": [P, ctypes.c_float, P, P, P, P, ctypes.c_int],
        "lz_batch_traverse": [P, ctypes.c_int, ctypes.c_float, ctypes.c_float, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p],
        "lz_batch_backpropagate": [P, ctypes.c_int, ctypes.c_float, c_f32p, c_f32p, c_f32p, P, c_i32p],
        "lz_roots_get_distributions": [P, c_i32p, c_i32p],
        "lz_roots_get_values": [P, c_f32p],
        "lz_roots_get_trajectories": [P, c_i32p, ctypes.c_int],
        "lz_roots_get_minmax": [P, c_f32p],
        "lz_sroots_create": [P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(P)],
        "lz_sroots_create_discrete": [P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(P)],
        "lz_sroots_prepare": [P, ctypes.c_float, P, c_f32p, c_f32p, c_i32p, P],
        "lz_sbatch_traverse": [P, ctypes.c_int, ctypes.c_float, ctypes.c_float, c_i32p, c_i32p, c_i32p, c_f32p, c_i32p],
        "lz_sbatch_backpropagate": [P, ctypes.c_int, ctypes.c_float, c_f32p, c_f32p, c_f32p, c_i32p, c_i32p, P],
        "lz_sroots_get_distributions": [P, c_i32p],
        "lz_sroots_get_sampled_actions": [P, c_f32p],
        "lz_sroots_set_given": [P, P, ctypes.c_int],
        "lz_roots_get_search_results": [P, c_i32p, c_i32p, c_f32p, P, P],
        "lz_roots_get_search_results_select": [P, c_i32p, c_i32p, c_f32p, P, P, ctypes.c_double, ctypes.c_int, ctypes.c_uint64, c_i32p, P],
        "lz_groots_prepare": [P, ctypes.c_float, P, c_f32p, c_f32p, c_f32p, c_i32p],
        "lz_gbatch_traverse": [P, ctypes.c_int, ctypes.c_int, ctypes.c_float, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p],
        "lz_gbatch_back_propagate": [P, ctypes.c_int, ctypes.c_float, c_f32p, c_f32p, c_f32p],
        "lz_groots_get_policies": [P, ctypes.c_float, P, P],
        "lz_gsearch": [P, ctypes.c_int, ctypes.c_int, ctypes.c_float],
        "lz_search_with_reuse": [P, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_float, c_i32p, c_f32p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)],
        "lz_batch_traverse_with_reuse": [P, ctypes.c_int, ctypes.c_float, ctypes.c_float, c_i32p, c_i32p, c_f32p, c_i32p, c_i32p, c_i32p, c_i32p],
        "lz_batch_backpropagate_with_reuse": [P, ctypes.c_int, ctypes.c_float, P, P, P, ctypes.c_int, P, c_i32p, c_i32p, c_i32p, c_f32p],
        "lz_roots_select_action": [P, ctypes.c_double, ctypes.c_int, ctypes.c_uint64, c_i32p, np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")],
        "lz_model_create": [P, ctypes.POINTER(ModelCfg)],
        "lz_model_set_tensor": [P, ctypes.c_char_p, c_f32p, ctypes.POINTER(ctypes.c_int64), ctypes.c_int],
        "lz_model_finalize": [P],
        "lz_initial_inference": [P, P],
        "lz_initial_inference_host": [P, c_f32p],
        "lz_roots_get_root_outputs": [P, c_f32p, c_f32p],
        "lz_roots_prepare_from_inference": [P, ctypes.c_float, P, c_i32p],
        "lz_search": [P, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_float],
        "lz_profile_enable": [P, ctypes.c_int],
        "lz_profile_read": [P, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double)],
        "lz_roots_enable_trace": [P, ctypes.c_int],
        "lz_roots_read_trace": [P, ctypes.c_int, c_i32p],
        "lz_roots_read_sim_outputs": [P, ctypes.c_int, c_f32p, c_f32p, c_f32p],
        "lz_roots_read_latent": [P, ctypes.c_int, c_f32p],
        "lz_roots_read_hidden": [P, ctypes.c_int, c_f32p, c_f32p],
        "lz_roots_read_debug_logits": [P, ctypes.c_int, c_f32p],
    }
    for name, argtypes in sig.items():
        getattr(L, name).argtypes = argtypes
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise LzError("liblz_mi355: %s (status %d)" % (lib().lz_last_error().decode(), rc))


_engines = {}


def default_engine(device_index=None):
    """One engine (HIP stream + weights) per device per process; device defaults to LOCAL_RANK."""
    if device_index is None:
        device_index = int(os.environ.get("LOCAL_RANK", "0"))
    if device_index not in _engines:
        h = P()
        check(lib().lz_engine_create(device_index, ctypes.byref(h)))
        _engines[device_index] = h
    return _engines[device_index]


def f32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32))


def i32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.int32))
