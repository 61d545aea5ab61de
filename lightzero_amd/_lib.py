"""ctypes binding of liblz_mi355.so (include/lz_mi355.h).  There is NO fallback: if the HIP
library is missing or no gfx950 device is visible, compute entry points raise."""
import ctypes
import os
import sys

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# LZ_MI355_LIB: tools/ point this at liblz_mi355_dbg.so (the -DLZ_DEBUG_KNOBS build); everything else loads the release library
LIB_PATH = os.environ.get("LZ_MI355_LIB") or os.path.join(_PKG, "liblz_mi355.so")


class LzError(RuntimeError):
    pass


class ModelCfg(ctypes.Structure):
    _fields_ = [("model_type", ctypes.c_int), ("obs_c", ctypes.c_int), ("obs_h", ctypes.c_int), ("obs_w", ctypes.c_int),
                ("action_space_size", ctypes.c_int), ("num_channels", ctypes.c_int), ("lstm_hidden_size", ctypes.c_int),
                ("head_channels", ctypes.c_int), ("head_hidden", ctypes.c_int), ("support_size", ctypes.c_int),
                ("support_min", ctypes.c_float), ("bn_eps", ctypes.c_float), ("downsample", ctypes.c_int),
                ("activation", ctypes.c_int), ("res_connection_in_dynamics", ctypes.c_int), ("action_encoding", ctypes.c_int),
                ("num_of_sampled_actions", ctypes.c_int), ("sigma_type", ctypes.c_int), ("bound_type", ctypes.c_int),
                ("ln_eps", ctypes.c_float), ("num_res_blocks", ctypes.c_int), ("reward_support_size", ctypes.c_int),
                ("reward_support_min", ctypes.c_float), ("precision", ctypes.c_int), ("state_norm", ctypes.c_int), ("scalar_heads", ctypes.c_int)]


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
    L.lz_engine_model_uid.restype = ctypes.c_uint64
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
        "lz_roots_reseed": [P, ctypes.c_uint64],
        "lz_roots_prepare": [P, ctypes.c_float, P, c_f32p, c_f32p, c_i32p],
        "lz_roots_prepare_device": [P, ctypes.c_float, P, P, P, P, ctypes.c_int],
        "lz_batch_traverse": [P, ctypes.c_int, ctypes.c_float, ctypes.c_float, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p],
        "lz_batch_backpropagate": [P, ctypes.c_int, ctypes.c_float, c_f32p, c_f32p, c_f32p, P, c_i32p],
        "lz_roots_get_distributions": [P, c_i32p, c_i32p],
        "lz_roots_get_values": [P, c_f32p],
        "lz_roots_get_trajectories": [P, c_i32p, ctypes.c_int],
        "lz_roots_get_minmax": [P, c_f32p],
        "lz_roots_get_root_priors": [P, c_f32p],
        "lz_sroots_create": [P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(P)],
        "lz_sroots_create_discrete": [P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(P)],
        "lz_sroots_prepare": [P, ctypes.c_float, P, c_f32p, c_f32p, c_i32p, P],
        "lz_sbatch_traverse": [P, ctypes.c_int, ctypes.c_float, ctypes.c_float, c_i32p, c_i32p, c_i32p, c_f32p, c_i32p],
        "lz_sbatch_backpropagate": [P, ctypes.c_int, ctypes.c_float, c_f32p, c_f32p, c_f32p, c_i32p, c_i32p, P],
        "lz_sroots_get_distributions": [P, c_i32p],
        "lz_sroots_get_sampled_actions": [P, c_f32p],
        "lz_sroots_get_node_actions": [P, ctypes.c_int, c_f32p],
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
        "lz_model_set_tensor_device": [P, ctypes.c_char_p, P, ctypes.POINTER(ctypes.c_int64), ctypes.c_int],
        "lz_model_finalize": [P],
        "lz_model_flat_layout": [P, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)],
        "lz_model_flat_entry": [P, ctypes.c_int64, ctypes.c_char_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)],
        "lz_model_flat_host_buffer": [P, ctypes.POINTER(ctypes.POINTER(ctypes.c_float))],
        "lz_model_refresh_flat": [P, P, ctypes.c_int64, ctypes.c_int],
        "lz_model_weights_digest": [P, ctypes.POINTER(ctypes.c_uint64)],
        "lz_initial_inference": [P, P],
        "lz_initial_inference_host": [P, c_f32p],
        "lz_roots_get_root_outputs": [P, c_f32p, c_f32p],
        "lz_roots_adopt_inference": [P, P],
        "lz_roots_prepare_from_inference": [P, ctypes.c_float, P, c_i32p],
        "lz_roots_prepare_from_inference_dirichlet": [P, ctypes.c_float, ctypes.c_float, c_i32p],
        "lz_search": [P, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_float],
        "lz_profile_enable": [P, ctypes.c_int],
        "lz_profile_read": [P, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double)],
        "lz_roots_enable_trace": [P, ctypes.c_int],
        "lz_roots_read_trace": [P, ctypes.c_int, c_i32p],
        "lz_roots_read_sim_outputs": [P, ctypes.c_int, c_f32p, c_f32p, c_f32p],
        "lz_roots_get_node_depths": [P, ctypes.c_int, c_i32p],
        "lz_roots_read_latent": [P, ctypes.c_int, c_f32p],
        "lz_roots_read_hidden": [P, ctypes.c_int, c_f32p, c_f32p],
        "lz_roots_read_debug_logits": [P, ctypes.c_int, c_f32p],
        "lz_roots_read_head_debug": [P, ctypes.c_int, ctypes.c_int, P, P],
        "lz_roots_enable_stamps": [P, ctypes.c_int],
        "lz_roots_read_stamps": [P, ctypes.c_int, np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")],
        "lz_debug_inverse_scalar_transform": [P, ctypes.c_int, c_f32p, ctypes.c_int64, c_f32p],
        "lz_roots_write_latent": [P, ctypes.c_int, c_f32p],
        "lz_roots_write_hidden": [P, ctypes.c_int, c_f32p, c_f32p],
        "lz_recurrent_inference": [P, c_i32p, P, P, P, ctypes.c_int, ctypes.c_int],
        "lz_engine_model_uid": [P],
        "lz_rows_width": [ctypes.c_int, ctypes.c_int],
        "lz_wino_weights": [c_f32p, ctypes.c_int, ctypes.c_int, c_f32p],
        "lz_rows_extra_words": [P],
        "lz_roots_collect_rows_ex": [P, ctypes.c_double, ctypes.c_int, ctypes.c_uint64, ctypes.c_float, P, ctypes.c_int, P, P, ctypes.c_int, c_f32p, P],
        "lz_roots_collect_rows_begin": [P, ctypes.c_double, ctypes.c_int, ctypes.c_uint64, ctypes.c_float, P, ctypes.c_int, P, P, ctypes.c_int, ctypes.c_int],
        "lz_roots_collect_rows_end": [P, c_f32p, P],
        "lz_roots_collect_rows": [P, ctypes.c_double, ctypes.c_int, ctypes.c_uint64, P, ctypes.c_int, P, P, ctypes.c_int, c_f32p, P],
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


_engine_death_hooks = []   # callables(engine address): whoever parks device handles of an engine (the roots-handle cache) frees them first


class OwnedEngine(ctypes.c_void_p):
    """An engine created by ``new_engine``: destroyed (weights, stream, workspace back to the device) when the last Python object
    that holds it -- the model, every Roots built on it -- is gone.  The per-device default engine is a plain pointer and lives as
    long as the process."""

    def __del__(self):
        try:
            if self.value:
                for hook in _engine_death_hooks:
                    hook(self.value)
                lib().lz_engine_destroy(self)
                self.value = None
        except Exception:   # interpreter shutdown: the library may be gone already
            pass


def new_engine(device_index=None):
    """a fresh engine (its own HIP stream, room for one model) on the device; freed with its last user (OwnedEngine)"""
    if device_index is None:
        device_index = int(os.environ.get("LOCAL_RANK", "0"))
    h = OwnedEngine()
    check(lib().lz_engine_create(device_index, ctypes.byref(h)))
    return h


def engine_for_new_model(device_index=None):
    """An engine holds ONE model (lz_model_create replaces the previous one).  The first model of a process gets the default
    engine, every further one its own engine, so that two live model objects never share -- and silently overwrite -- weights."""
    e = default_engine(device_index)
    if lib().lz_engine_model_uid(e) == 0:
        return e
    return new_engine(device_index)


_seed_counter = [0]


def process_seed():
    """Default seed of the device-side random streams (stochastic tie-breaks, sampled actions, select_action): derived from
    np.random's state WITHOUT consuming it -- so ``np.random.seed`` / the config seed govern it like they govern the reference's
    Dirichlet noise, and the noise stream itself does not shift with the number of Roots objects a run happens to build -- mixed
    with a per-process counter (every Roots object gets its own stream) and the rank (data-parallel collectors do not explore in
    lock-step)."""
    import hashlib
    rank = int(os.environ.get("RANK", os.environ.get("LOCAL_RANK", "0")))
    # the calling thread's own stream when it has installed one (random_source: the pipelined collector's env groups) -- the global
    # stream is advanced by whichever thread steps the other group's envs, so hashing IT would make the seed depend on thread timing
    st = rs().get_state()
    h = hashlib.blake2b(digest_size=8)
    h.update(np.asarray(st[1]).tobytes())
    h.update(np.asarray([st[2], _seed_counter[0]], np.int64).tobytes())
    _seed_counter[0] += 1
    return (int.from_bytes(h.digest(), "little") ^ ((rank + 1) * 0x9E3779B97F4A7C15)) & (2 ** 63 - 1)


_tls = None


def rs():
    """The host-side random source of the policies / tree shims (Dirichlet root noise, eps-greedy draws, seeds of the device-side
    streams): the global ``np.random`` -- what the reference uses, governed by ``np.random.seed`` -- unless the calling THREAD has
    installed a ``np.random.RandomState`` of its own with ``random_source``.  The vectorised collector does that for its pipelined env
    groups: a group's policy forward runs on a worker thread while the main thread steps the other group's envs, and two threads
    interleaving on the one global stream would make seeded runs irreproducible."""
    src = getattr(_tls, "rs", None) if _tls is not None else None
    return src if src is not None else np.random


class random_source(object):
    """``with random_source(np.random.RandomState(seed)):`` -- this thread's draws come from that stream inside the block"""

    def __init__(self, state):
        self.state = state

    def __enter__(self):
        global _tls
        if _tls is None:
            import threading
            _tls = threading.local()
        self.prev = getattr(_tls, "rs", None)
        _tls.rs = self.state
        return self.state

    def __exit__(self, *exc):
        _tls.rs = self.prev
        return False


def f32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32))


def i32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.int32))
