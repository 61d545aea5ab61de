"""The fused search must not depend on (a) what the allocator hands back (LZ_POISON fills every fresh device allocation of
the library with a byte pattern) or (b) work other libraries have queued on the null stream while the engine sets up its
trees on its own non-blocking stream (a null-stream memset once landed after the first prepare).  Bit-identical root
values and visit counts in every arm."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _search(B=64, A=6, S=20, seed=5, before=None):
    import torch
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    from lightzero_amd.model.synthetic import efficientzero_state_dict
    model = EfficientZeroModel(action_space_size=A).load_state_dict(efficientzero_state_dict(seed=0, action_space_size=A))
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(seed)).cuda().contiguous()
    torch.cuda.synchronize()
    if before is not None:
        before()
    roots = ez_tree.Roots(B, [list(range(A))] * B, action_space_size=A, max_simulations=S)
    roots.set_tiebreak(0)
    model.initial_inference(obs, roots, fetch=False)
    roots.prepare_from_inference_no_noise([-1] * B)
    L.check(L.lib().lz_search(roots._h, S, 19652, 1.25, 0.997, 5, 0.01))
    return np.asarray(roots.get_values(), np.float32), np.asarray(roots.get_distributions(), np.int32)


def test_search_independent_of_fresh_allocation_contents():
    old = os.environ.get("LZ_POISON")
    try:
        os.environ.pop("LZ_POISON", None)
        v0, d0 = _search()
        for byte in ("0x7f", "0xff", "0x00"):
            os.environ["LZ_POISON"] = byte
            v, d = _search()
            assert np.array_equal(d, d0), byte
            assert np.array_equal(v.view(np.uint32), v0.view(np.uint32)), byte
    finally:
        if old is None:
            os.environ.pop("LZ_POISON", None)
        else:
            os.environ["LZ_POISON"] = old
    assert (d0.sum(1) == 20).all()


def test_search_independent_of_pending_null_stream_work():
    import torch
    v0, d0 = _search()
    keep = []

    def busy():  # ~10 ms of default-stream work still running while the engine allocates and prepares its roots
        for _ in range(6):
            keep.append(torch.randn(256 * 1024 * 1024, device="cuda"))
    v, d = _search(before=busy)
    torch.cuda.synchronize()
    assert np.array_equal(d, d0)
    assert np.array_equal(v.view(np.uint32), v0.view(np.uint32))
