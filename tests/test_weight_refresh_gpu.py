"""GPU: weight refresh from DEVICE tensors (lz_model_set_tensor_device; shard.broadcast_state_dict(..., on_device=True) hands views of the
broadcast buffer to model.load_state_dict): the refreshed model must be bit-identical to a model loaded from the same weights as numpy
arrays, captured search graphs stay valid, and a model refreshed twice keeps its device buffers."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _outputs(model, obs, A, S=6):
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    B = obs.shape[0]
    roots = ez_tree.Roots(B, [list(range(A))] * B, action_space_size=A, max_simulations=S, engine=model.engine)
    roots.set_tiebreak(0)
    roots._ensure(A)
    L.check(L.lib().lz_initial_inference(roots._h, obs.data_ptr()))
    roots.prepare_from_inference_no_noise([-1] * B)
    L.check(L.lib().lz_search(roots._h, S, 19652, 1.25, 0.997, 5, 0.01))
    val = np.zeros(B, np.float32); pol = np.zeros((B, A), np.float32)
    L.check(L.lib().lz_roots_get_root_outputs(roots._h, val, pol.reshape(-1)))
    return np.array(roots.get_distributions()), np.array(roots.get_values(), np.float32), val, pol, roots


def test_refresh_from_device_tensors_equals_refresh_from_host_arrays():
    from oracle import torch_models as tm
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    A, B = 6, 24
    sd0 = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A), seed=1).state_dict()
    sd1 = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A), seed=2).state_dict()
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(4)).cuda().contiguous()
    torch.cuda.synchronize()
    host = EfficientZeroModel(action_space_size=A).load_state_dict(sd1)
    want = _outputs(host, obs, A)
    dev = EfficientZeroModel(action_space_size=A).load_state_dict(sd0)
    before = _outputs(dev, obs, A)
    assert not np.array_equal(before[3], want[3])
    # the learner's new weights arrive as views of ONE flat device buffer (what shard.broadcast_state_dict(on_device=True) returns)
    names = sorted(k for k in sd1 if not k.endswith("num_batches_tracked"))
    flat = torch.cat([sd1[k].reshape(-1).float() for k in names]).cuda()
    views, off = {}, 0
    for k in names:
        n = sd1[k].numel()
        views[k] = flat[off:off + n].view(*sd1[k].shape)
        off += n
    dev.load_state_dict(views)
    got = _outputs(dev, obs, A)
    for a, b in zip(got[:4], want[:4]):
        assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
    # a second refresh (back to the first weights) on live roots: the captured graph of `before`'s roots is replayed on the new weights
    dev.load_state_dict({k: v.cuda() for k, v in sd0.items() if not k.endswith("num_batches_tracked")})
    again = _outputs(dev, obs, A)
    for a, b in zip(again[:4], before[:4]):
        assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
