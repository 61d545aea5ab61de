"""GPU: a LightZero checkpoint ({'model', 'target_model', 'optimizer'}, lzero/policy/muzero.py:1043-1061) loads into the engine model and
into a policy exactly like the bare state_dict of its online network; loading another one is a weight refresh."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _out(model, obs):
    o = model.initial_inference(obs)
    return np.asarray(o.policy_logits, np.float32), np.asarray(o.value, np.float32)


def test_checkpoint_layout_loads_like_the_bare_state_dict_and_refreshes():
    from lightzero_amd.model.synthetic import efficientzero_state_dict
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    from lightzero_amd.policy.efficientzero import EfficientZeroPolicy
    sd0, sd1 = (efficientzero_state_dict(seed=s, action_space_size=6) for s in (0, 1))
    obs = torch.rand(9, 4, 96, 96, generator=torch.Generator().manual_seed(2))
    bare0 = _out(EfficientZeroModel(action_space_size=6).load_state_dict(sd0), obs)
    bare1 = _out(EfficientZeroModel(action_space_size=6).load_state_dict(sd1), obs)
    assert not np.array_equal(bare0[0], bare1[0])

    def ckpt(sd, other):
        return {"model": {"module." + k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()},
                "target_model": {k: torch.as_tensor(np.asarray(v)) for k, v in other.items()},
                "optimizer": {"state": {}, "param_groups": [{"lr": 3e-3}]}, "last_iter": 1000}
    model = EfficientZeroModel(action_space_size=6).load_state_dict(ckpt(sd0, sd1))
    got = _out(model, obs)
    assert np.array_equal(got[0], bare0[0]) and np.array_equal(got[1], bare0[1])
    # the policy's checkpoint entry point: a weight refresh of the model it plays with
    pol = EfficientZeroPolicy(dict(num_simulations=4), model)
    pol._load_state_dict_learn(ckpt(sd1, sd0))
    got = _out(model, obs)
    assert np.array_equal(got[0], bare1[0]) and np.array_equal(got[1], bare1[1])


def test_conv_sampled_efficientzero_takes_a_whole_checkpoint():
    """SampledEfficientZeroModel.load_state_dict renames the reference's fc_value_head / fc_policy_head keys -- it has to do so
    INSIDE the checkpoint's 'model' dict, not on the checkpoint's top-level keys."""
    from oracle import torch_models as tm
    from lightzero_amd.model.sampled_efficientzero_model import SampledEfficientZeroModel
    kw = dict(observation_shape=(4, 64, 64), action_space_size=6, num_of_sampled_actions=5, norm_type='BN', downsample=True)
    ref = tm.synthetic_init(tm.SampledEfficientZeroModel(**kw), seed=3)
    sd = ref.state_dict()
    assert any(k.startswith("prediction_network.fc_value_head.") for k in sd)
    obs = torch.rand(5, 4, 64, 64, generator=torch.Generator().manual_seed(4))
    bare = _out(SampledEfficientZeroModel(**kw).load_state_dict(sd), obs)
    ckpt = {"model": {"module." + k: v.clone() for k, v in sd.items()}, "target_model": {k: v.clone() for k, v in sd.items()},
            "optimizer": {"state": {}, "param_groups": [{"lr": 3e-3}]}, "last_iter": 10}
    got = _out(SampledEfficientZeroModel(**kw).load_state_dict(ckpt), obs)
    assert np.array_equal(got[0], bare[0]) and np.array_equal(got[1], bare[1])
