"""GPU: the C ABI's error behaviour -- wrong arguments must come back as a status < 0 with a message (lz_last_error), never as a crash,
a hang or a silently wrong result; the handle stays usable afterwards.  Calls go through ctypes directly (no Python-side checks)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _err(rc):
    from lightzero_amd import _lib as L
    assert rc < 0, "expected an error status, got %d" % rc
    msg = L.lib().lz_last_error().decode()
    assert len(msg) > 8, msg
    return msg


def test_bad_arguments_are_refused_and_the_handle_survives():
    from oracle import torch_models as tm
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    lib = L.lib()
    A, B, S = 6, 8, 5
    model = EfficientZeroModel(action_space_size=A).load_state_dict(tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A), seed=1).state_dict())
    eng = model.engine
    h = L.P()
    cnt, flat = L.i32([A] * B), L.i32(list(range(A)) * B)
    # ---- creation
    _err(lib.lz_roots_create(eng, 0, 0, A, S, flat, cnt, ctypes.byref(h)))            # no roots
    _err(lib.lz_roots_create(eng, 0, B, 0, S, flat, cnt, ctypes.byref(h)))            # no actions
    _err(lib.lz_roots_create(eng, 0, B, A, -1, flat, cnt, ctypes.byref(h)))           # negative simulations
    _err(lib.lz_roots_create(eng, 99, B, A, S, flat, cnt, ctypes.byref(h)))           # unknown tree variant
    _err(lib.lz_roots_create(eng, 0, B, A, S, L.i32([A + 3] * (A * B)), cnt, ctypes.byref(h)))   # legal action out of range
    _err(lib.lz_roots_create(eng, 0, B, A, S, flat, L.i32([A + 1] * B), ctypes.byref(h)))        # more legal actions than actions
    # ---- a good handle, then bad calls on it
    roots = ez_tree.Roots(B, [list(range(A))] * B, action_space_size=A, max_simulations=S, engine=eng)
    roots._ensure(A)
    obs = torch.rand(B, 4, 96, 96, device="cuda")
    _err(lib.lz_search(roots._h, S, 19652, 1.25, 0.997, 5, 0.01))                     # search before any inference / prepare
    L.check(lib.lz_initial_inference(roots._h, obs.data_ptr()))
    _err(lib.lz_search(roots._h, S, 19652, 1.25, 0.997, 5, 0.01))                     # inference but no prepare
    L.check(lib.lz_roots_prepare_from_inference(roots._h, 0.25, None, L.i32([-1] * B)))
    _err(lib.lz_search(roots._h, S + 1, 19652, 1.25, 0.997, 5, 0.01))                 # more simulations than the pools hold
    _err(lib.lz_search(roots._h, 0, 19652, 1.25, 0.997, 5, 0.01))                     # zero simulations
    _err(lib.lz_search(None, S, 19652, 1.25, 0.997, 5, 0.01))                         # NULL handle
    _err(lib.lz_initial_inference(roots._h, None))                                   # NULL observation
    out = np.zeros(B, np.float32)
    _err(lib.lz_roots_read_sim_outputs(roots._h, S + 5, out, out, np.zeros(B * A, np.float32)))   # slot out of range
    bad_actions = L.i32([A + 2] * B)
    _err(lib.lz_recurrent_inference(roots._h, L.i32([0] * B), bad_actions.ctypes.data, None, None, 0, 1))   # action out of range
    _err(lib.lz_recurrent_inference(roots._h, L.i32([S + 9] * B), L.i32([0] * B).ctypes.data, None, None, 0, 1))   # parent slot out of range
    # ---- the handle still works
    L.check(lib.lz_search(roots._h, S, 19652, 1.25, 0.997, 5, 0.01))
    assert all(sum(d) == S for d in roots.get_distributions())


def test_destroying_a_roots_handle_twice_or_after_its_engine_is_an_error_not_a_crash():
    """a host-language binding frees handles from finalizers whose order it does not control: lz_roots_destroy accepts live handles
    only, lz_engine_destroy takes the engine's remaining roots with it"""
    from lightzero_amd import _lib as L
    lib = L.lib()
    A, B, S = 4, 3, 5
    eng = L.P()
    L.check(lib.lz_engine_create(0, ctypes.byref(eng)))
    cnt, flat = L.i32([A] * B), L.i32(list(range(A)) * B)
    h1, h2 = L.P(), L.P()
    L.check(lib.lz_roots_create(eng, 0, B, A, S, flat, cnt, ctypes.byref(h1)))
    L.check(lib.lz_roots_create(eng, 1, B, A, S, flat, cnt, ctypes.byref(h2)))
    L.check(lib.lz_roots_destroy(h1))
    assert "not a live roots handle" in _err(lib.lz_roots_destroy(h1))     # twice
    L.check(lib.lz_engine_destroy(eng))                                       # ... takes h2 with it
    assert "not a live roots handle" in _err(lib.lz_roots_destroy(h2))     # after its engine
