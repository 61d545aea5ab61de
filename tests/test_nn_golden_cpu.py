"""The committed NN goldens (tests/golden/nn_*.npz = outputs of the REFERENCE's own model modules, made by
tests/golden/make_golden_nn.py) against the oracle restatement (oracle/torch_models.py), on any machine: weights digest and
every stored output bit-equal.  This is the form of the NN pin that travels to boxes without /root/reference."""
import os
import sys

import numpy as np
import pytest
import torch

import nn_cases
from oracle import torch_models as tm

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
from make_golden_nn import weights_digest  # noqa: E402


@pytest.mark.parametrize("name", sorted(nn_cases.CASES))
def test_oracle_reproduces_reference_goldens(name):
    case = nn_cases.CASES[name]
    fam, kw = case["family"], case["kw"]
    g = np.load(os.path.join(GOLD, "nn_%s.npz" % name))
    ora = tm.synthetic_init(nn_cases.oracle_class(tm, fam)(**kw), seed=case["seed"])
    assert weights_digest(ora.state_dict()) == bytes(g["weights_sha256"]).decode(), "seeded weights differ from the golden's"
    cat = bool(kw.get("categorical_distribution", True))   # False: one-output heads, h^-1 on the scalar itself (scaling_transform.py:88-89)
    ist = tm.InverseScalarTransform(kw.get("value_support_range", (-300., 301., 1.)), cat)
    rist = tm.InverseScalarTransform(kw.get("reward_support_range", kw.get("value_support_range", (-300., 301., 1.))), cat)
    obs, actions = nn_cases.inputs(case)

    def eq(a, key):
        a = a.numpy() if hasattr(a, "numpy") else np.asarray(a)
        assert a.shape == g[key].shape and np.array_equal(a, g[key]), "%s/%s: max |d| = %g" % (name, key, np.abs(a - g[key]).max())

    with torch.no_grad():
        o = ora.initial_inference(torch.from_numpy(obs))
        eq(o.latent_state, "init_latent"); eq(o.value, "init_value_logits"); eq(o.policy_logits, "init_policy")
        eq(ist(o.value.clone()).reshape(-1), "init_value")
        for s in range(nn_cases.STEPS):
            a = torch.from_numpy(actions[s])
            lat = torch.from_numpy(g["s%d_in_latent" % s])
            if nn_cases.has_lstm(fam):
                hc = (torch.from_numpy(g["s%d_in_h" % s]).unsqueeze(0), torch.from_numpy(g["s%d_in_c" % s]).unsqueeze(0))
                o = ora.recurrent_inference(lat, hc, a)
                eq(o.reward_hidden_state[0][0], "s%d_h" % s); eq(o.reward_hidden_state[1][0], "s%d_c" % s)
                rew = o.value_prefix
            else:
                o = ora.recurrent_inference(lat, a)
                rew = o.reward
            eq(o.latent_state, "s%d_latent" % s); eq(rew, "s%d_reward_logits" % s); eq(o.value, "s%d_value_logits" % s)
            eq(o.policy_logits, "s%d_policy" % s)
            eq(rist(rew.clone()).reshape(-1), "s%d_reward" % s); eq(ist(o.value.clone()).reshape(-1), "s%d_value" % s)
