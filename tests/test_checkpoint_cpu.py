"""CPU: the weight-ingest format of the path.  A LightZero checkpoint is the policy's learn-mode state
{'model', 'target_model', 'optimizer'} (lzero/policy/muzero.py:1043-1047); the engine models take it as torch.load returns it."""
import numpy as np
import pytest

from lightzero_amd.model.efficientzero_model import unwrap_checkpoint


def _sd(prefix=""):
    return {prefix + "representation_network.conv.weight": np.zeros((2, 3), np.float32), prefix + "prediction_network.fc_value.0.bias": np.ones(4, np.float32)}


def test_checkpoint_layouts_unwrap_to_the_online_network():
    bare = _sd()
    assert unwrap_checkpoint(bare).keys() == bare.keys()
    ckpt = {"model": _sd(), "target_model": {k: v + 1 for k, v in _sd().items()}, "optimizer": {"state": {}, "param_groups": []}, "last_iter": 7}
    out = unwrap_checkpoint(ckpt)
    assert out.keys() == bare.keys() and all((out[k] == bare[k]).all() for k in bare)
    tgt = unwrap_checkpoint(ckpt, which="target_model")
    assert all((tgt[k] == bare[k] + 1).all() for k in bare)
    # DistributedDataParallel / torch.compile wrappers around the learn model
    assert unwrap_checkpoint({"model": _sd("module.")}).keys() == bare.keys()
    assert unwrap_checkpoint(_sd("_orig_mod.module.")).keys() == bare.keys()


def test_malformed_checkpoints_are_refused_with_the_reason():
    with pytest.raises(KeyError, match="target_model"):
        unwrap_checkpoint({"model": _sd()}, which="target_model")
    with pytest.raises(TypeError, match="dict"):
        unwrap_checkpoint({"state": {"a": 1}})
    with pytest.raises(TypeError, match="mapping"):
        unwrap_checkpoint([1, 2, 3])


def test_policies_expose_the_reference_entry_points():
    from lightzero_amd.policy.efficientzero import EfficientZeroPolicy
    from lightzero_amd.policy.muzero import MuZeroPolicy
    from lightzero_amd.policy.gumbel_muzero import GumbelMuZeroPolicy
    from lightzero_amd.policy.sampled_efficientzero import SampledEfficientZeroPolicy
    for P in (EfficientZeroPolicy, MuZeroPolicy, GumbelMuZeroPolicy, SampledEfficientZeroPolicy):
        for name in ("_load_state_dict_learn", "_load_state_dict_collect", "_load_state_dict_eval", "load_state_dict"):
            assert callable(getattr(P, name)), (P, name)

    class M(object):
        def __init__(self):
            self.got = []

        def load_state_dict(self, sd):
            self.got.append(sd)
    p = EfficientZeroPolicy.__new__(EfficientZeroPolicy)
    p._collect_model = p._eval_model = M()
    p._load_state_dict_learn({"model": _sd()})
    assert len(p._collect_model.got) == 1      # one model object behind both modes: loaded once
