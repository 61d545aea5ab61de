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


def _digest(model):
    import ctypes
    from lightzero_amd import _lib as L
    h = ctypes.c_uint64(0)
    L.check(L.lib().lz_model_weights_digest(model.engine, ctypes.byref(h)))
    return h.value


def _families():
    from oracle import torch_models as tm
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    from lightzero_amd.model.muzero_model import MuZeroModel
    from lightzero_amd.model.sampled_efficientzero_model import SampledEfficientZeroModel
    return {
        "ez_atari96": (EfficientZeroModel, tm.EfficientZeroModel, dict(action_space_size=6)),
        "ez_atari96_not_one_hot": (EfficientZeroModel, tm.EfficientZeroModel, dict(action_space_size=6, discrete_action_encoding_type='not_one_hot')),
        "ez_atari64_2blocks": (EfficientZeroModel, tm.EfficientZeroModel, dict(action_space_size=18, observation_shape=(4, 64, 64), num_res_blocks=2)),
        "mz_breakout": (MuZeroModel, tm.MuZeroModel, dict(action_space_size=4)),
        "mz_go9": (MuZeroModel, tm.MuZeroModel, dict(action_space_size=82, observation_shape=(17, 9, 9), downsample=False)),
        "mz_gomoku32": (MuZeroModel, tm.MuZeroModel, dict(action_space_size=36, observation_shape=(3, 6, 6), downsample=False, num_channels=32,
                                                       reward_support_range=(-10., 11., 1.), value_support_range=(-10., 11., 1.))),
        "sez_atari64": (SampledEfficientZeroModel, tm.SampledEfficientZeroModel,
                        dict(observation_shape=(4, 64, 64), action_space_size=6, num_of_sampled_actions=5, norm_type='BN', downsample=True)),
    }


@pytest.mark.parametrize("family", ["ez_atari96", "ez_atari96_not_one_hot", "ez_atari64_2blocks", "mz_breakout", "mz_go9", "mz_gomoku32", "sez_atari64"])
def test_device_side_refresh_leaves_the_bytes_a_host_finalize_leaves(family):
    """lz_model_refresh_flat re-lays the weights out with kernels (gather maps recorded from the host packers themselves;
    BatchNorm folding, Winograd G g G^T in binary64, action table, LSTM bias as derived tensors).  After a refresh -- from host arrays,
    from consecutive views of one device buffer, from scattered device tensors -- EVERY device weight buffer must hold exactly the bytes
    lz_model_set_tensor + lz_model_finalize leave for the same state_dict (FNV-1a over all buffers), for every convolutional family."""
    from oracle import torch_models as tm
    M, R, kw = _families()[family]
    sd0 = tm.synthetic_init(R(**kw), seed=1).state_dict()
    sd1 = tm.synthetic_init(R(**kw), seed=2).state_dict()
    want = _digest(M(**kw).load_state_dict(sd1))
    model = M(**kw).load_state_dict(sd0)
    d0 = _digest(model)
    assert d0 != want
    model.load_state_dict(sd1)                                        # host arrays -> pinned staging -> kernels
    assert model._flat_layout not in (None, False), "the device-side refresh was not taken"
    assert _digest(model) == want
    model.load_state_dict(sd0)
    assert _digest(model) == d0
    names = sorted(k for k in sd1 if not k.endswith("num_batches_tracked"))
    flat = torch.cat([sd1[k].reshape(-1).float() for k in names]).cuda()
    views, off = {}, 0
    for k in names:
        n = sd1[k].numel()
        views[k] = flat[off:off + n].view(*sd1[k].shape)
        off += n
    model.load_state_dict(views)                                      # consecutive views of one device buffer: passed by pointer
    assert _digest(model) == want
    model.load_state_dict({k: sd0[k].cuda() for k in names})          # scattered device tensors: one torch.cat
    assert _digest(model) == d0
    from lightzero_amd import shard
    fsd = shard.flat_state_dict(sd1, "cuda")                          # shard.FlatStateDict: the buffer goes over by pointer, no walk
    if family != "sez_atari64":   # (that model renames the reference's head keys first: its buffer order is not the reference names' order)
        assert [n for n, _, _ in fsd.layout] == [n for n, _, _ in model._flat_layout[0]]
    model.load_state_dict(fsd)
    assert _digest(model) == want


def test_host_path_after_device_refreshes_sees_the_refreshed_weights():
    """after device-side refreshes the library's host copy of the state_dict is stale; a later per-tensor lz_model_set_tensor (the host
    path: e.g. ONE tensor patched) first restores it from the device, so the finalize that follows re-lays out what is really loaded"""
    import ctypes
    from oracle import torch_models as tm
    from lightzero_amd import _lib as L
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    A = 6
    sd0 = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A), seed=1).state_dict()
    sd1 = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A), seed=2).state_dict()
    key = "prediction_network.fc_policy.3.bias"
    mixed = dict(sd1)
    mixed[key] = sd0[key] + 1.0
    want = _digest(EfficientZeroModel(action_space_size=A).load_state_dict(mixed))
    model = EfficientZeroModel(action_space_size=A).load_state_dict(sd0)
    model.load_state_dict(sd1)     # device-side refresh
    arr = np.ascontiguousarray(mixed[key].numpy(), np.float32)
    shape = (ctypes.c_int64 * 1)(*arr.shape)
    L.check(L.lib().lz_model_set_tensor(model.engine, key.encode(), arr, shape, 1))
    L.check(L.lib().lz_model_finalize(model.engine))
    assert _digest(model) == want


def test_models_without_a_device_side_refresh_fall_back_to_the_host_path():
    """fast mode (bf16 fragments) and the MLP family re-lay their weights out on the host: load_state_dict still refreshes them"""
    from lightzero_amd.model.synthetic import efficientzero_state_dict
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    sd0, sd1 = (efficientzero_state_dict(seed=s, action_space_size=6) for s in (0, 1))
    want = _digest(EfficientZeroModel(action_space_size=6, fast_mode=True).load_state_dict(sd1))
    model = EfficientZeroModel(action_space_size=6, fast_mode=True).load_state_dict(sd0)
    model.load_state_dict(sd1)
    assert model._flat_layout is False
    assert _digest(model) == want


def test_refresh_is_stream_ordered_with_searches_on_live_roots():
    """a refresh enqueued between two searches of the same roots (no host synchronisation anywhere): the first search sees the old
    weights, the second the new ones -- both equal to searches on models that only ever held those weights"""
    from oracle import torch_models as tm
    from lightzero_amd import _lib as L
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    A, B = 6, 24
    sd0 = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A), seed=1).state_dict()
    sd1 = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A), seed=2).state_dict()
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(4)).cuda().contiguous()
    torch.cuda.synchronize()
    want0 = _outputs(EfficientZeroModel(action_space_size=A).load_state_dict(sd0), obs, A)
    want1 = _outputs(EfficientZeroModel(action_space_size=A).load_state_dict(sd1), obs, A)
    model = EfficientZeroModel(action_space_size=A).load_state_dict(sd0)
    _, _, _, _, roots = _outputs(model, obs, A)
    lib = L.lib()
    res = []
    for sd in (sd1, sd0, sd1):
        model.load_state_dict(sd)
        L.check(lib.lz_initial_inference(roots._h, obs.data_ptr()))
        roots.prepare_from_inference_no_noise([-1] * B)
        L.check(lib.lz_search(roots._h, 6, 19652, 1.25, 0.997, 5, 0.01))
        res.append((np.array(roots.get_distributions()), np.array(roots.get_values(), np.float32)))
    for got, want in zip(res, (want1, want0, want1)):
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
