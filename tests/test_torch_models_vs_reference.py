"""Pins the NN / h^-1 half of the oracle (oracle/torch_models.py) to the reference's OWN code: the reference model files are
imported from /root/reference as they lie (tests/ref_loader.py; DI-engine's ResBlock / MLP / ReparameterizationHead come from
tests/ref_stubs, the residual restatement), the oracle's seeded weights are loaded into them by name, and every output of
initial_inference and of a chain of recurrent_inference calls must be BIT-EQUAL.  lzero/policy/scaling_transform.py needs no
stub at all.  CPU only; skipped where /root/reference does not exist (the GPU box) -- there the committed
tests/golden/nn_*.npz (reference-module outputs, made by tests/golden/make_golden_nn.py) carry the pin."""
import numpy as np
import pytest
import torch

import nn_cases
import ref_loader
from oracle import torch_models as tm

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="/root/reference not present")


def _build(name):
    ref = ref_loader.load()
    case = nn_cases.CASES[name] if name in nn_cases.CASES else nn_cases.ORACLE_ONLY_CASES[name]
    fam = case["family"]
    ora = tm.synthetic_init(nn_cases.oracle_class(tm, fam)(**case["kw"]), seed=case["seed"])
    rmod = nn_cases.reference_class(ref, fam)(**nn_cases.reference_kwargs(case))
    res = rmod.load_state_dict(ora.state_dict(), strict=False)
    assert not res.unexpected_keys, res.unexpected_keys      # every oracle tensor exists under the same name in the reference
    assert not res.missing_keys, res.missing_keys            # and the reference's inference graph has no tensor the oracle lacks
    rmod.eval()
    return case, ora, rmod


def _eq(a, b, what):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert torch.equal(a, b), "%s differs: max |d| = %g" % (what, float((a.double() - b.double()).abs().max()))


@pytest.mark.parametrize("name", sorted(nn_cases.CASES) + sorted(nn_cases.ORACLE_ONLY_CASES))
def test_oracle_model_equals_reference_module(name):
    torch.manual_seed(0)
    case, ora, rmod = _build(name)
    fam = case["family"]
    obs, actions = nn_cases.inputs(case)
    obs = torch.from_numpy(obs)
    with torch.no_grad():
        o, r = ora.initial_inference(obs), rmod.initial_inference(obs)
        _eq(o.latent_state, r.latent_state, "initial latent_state")
        _eq(o.value, r.value, "initial value logits")
        _eq(o.policy_logits, r.policy_logits, "initial policy_logits")
        lat_o, lat_r = o.latent_state, r.latent_state
        if nn_cases.has_lstm(fam):
            hc_o, hc_r = o.reward_hidden_state, r.reward_hidden_state
            _eq(hc_o[0], hc_r[0], "initial h"); _eq(hc_o[1], hc_r[1], "initial c")
        for s in range(nn_cases.STEPS):
            a = torch.from_numpy(actions[s])
            if nn_cases.has_lstm(fam):
                o, r = ora.recurrent_inference(lat_o, hc_o, a), rmod.recurrent_inference(lat_r, hc_r, a)
                _eq(o.value_prefix, r.value_prefix, "step %d value_prefix logits" % s)
                _eq(o.reward_hidden_state[0], r.reward_hidden_state[0], "step %d h" % s)
                _eq(o.reward_hidden_state[1], r.reward_hidden_state[1], "step %d c" % s)
                hc_o, hc_r = o.reward_hidden_state, r.reward_hidden_state
            else:
                o, r = ora.recurrent_inference(lat_o, a), rmod.recurrent_inference(lat_r, a)
                _eq(o.reward, r.reward, "step %d reward logits" % s)
            _eq(o.latent_state, r.latent_state, "step %d latent_state" % s)
            _eq(o.value, r.value, "step %d value logits" % s)
            _eq(o.policy_logits, r.policy_logits, "step %d policy_logits" % s)
            lat_o, lat_r = o.latent_state, r.latent_state


def test_state_dict_names_and_shapes_equal_reference():
    """the weight-ingest format: same tensor names and shapes as the reference's inference graph (muzero.py:1043-1047 `model`)"""
    for name in nn_cases.CASES:
        case, ora, rmod = _build(name)
        a = {k: tuple(v.shape) for k, v in ora.state_dict().items()}
        b = {k: tuple(v.shape) for k, v in rmod.state_dict().items()}
        assert a == b, (name, set(a) ^ set(b))


@pytest.mark.parametrize("support", [(-300., 301., 1.), (-50., 51., 1.)])
def test_inverse_scalar_transform_equals_reference(support):
    ref = ref_loader.load()
    g = torch.Generator().manual_seed(5)
    n = len(torch.arange(*support))
    logits = torch.randn(4096, n, generator=g) * 3.0
    logits[:64] *= 10.0  # some nearly one-hot rows: large |value|
    theirs = ref.scaling_transform.InverseScalarTransform(ref.scaling_transform.DiscreteSupport(*support), True)
    ours = tm.InverseScalarTransform(support)
    _eq(ours(logits.clone()), theirs(logits.clone()), "InverseScalarTransform")


def test_inverse_scalar_transform_reference_own_test():
    """lzero/policy/tests/test_scaling_transform.py:7-19, the reference's own (and only) value-level check of h^-1: the function
    form and the handle agree on randn(16, 601) -- run here with the oracle's restatement as a third party, plus the analytic
    value of a one-hot distribution."""
    ref = ref_loader.load()
    st = ref.scaling_transform
    logit = torch.randn(16, 601, generator=torch.Generator().manual_seed(0))
    support = st.DiscreteSupport(-300., 301., 1.)
    output_1 = st.inverse_scalar_transform(logit, support)
    output_2 = st.InverseScalarTransform(support)(logit.clone())
    ours = tm.InverseScalarTransform()(logit.clone())
    assert output_1.shape == output_2.shape == ours.shape == (16, 1)
    assert (output_1 == output_2).all()
    assert torch.equal(ours, output_2)
    onehot = torch.full((3, 601), -1e4)
    onehot[:, 302] = 1e4  # support value 2
    x, eps = 2.0, 0.001
    expect = np.sign(x) * (((np.sqrt(1 + 4 * eps * (abs(x) + 1 + eps)) - 1) / (2 * eps)) ** 2 - 1)
    got = tm.InverseScalarTransform()(onehot)
    assert abs(float(got[0, 0]) - expect) < 1e-3 * (1 + abs(expect))
