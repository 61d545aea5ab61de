"""GPU: randomised sweep of the network kernels -- model configurations the committed goldens do not hold (action counts, residual
blocks, supports, observation sizes / board grids, channel widths, batch sizes that are not tile multiples) against the torch
restatement evaluated here (oracle/torch_models.py, bit-equal to the reference's own modules: tests/test_torch_models_vs_reference.py),
through the same teacher-forced checker as tests/test_nn_golden_gpu.py.

Bounds: north_star's 1e-5 (3e-4 after h^-1) per tensor class -- or, on a network whose own fp32 evaluation loses more than a third of that,
three times what torch's fp32 loses against a binary64 evaluation of the same inputs (three residual blocks unrolled three steps: the
latent's magnitude grows with every step and so does every fp32 implementation's error; the engine may differ from torch fp32 by the sum
of the two rounding errors).  A configuration the engine does not compile must be refused with a message, never computed wrongly."""
import copy

import numpy as np
import pytest
import torch

import nn_cases
import parity_record
from test_nn_golden_gpu import check_case

# LZ_FUZZ_SEED_OFFSET=n shifts every seeded sweep of this file to seeds n .. n + count - 1 (ad-hoc wider sweeps; the committed suite runs 0)
_OFF = int(__import__("os").environ.get("LZ_FUZZ_SEED_OFFSET", "0"))

pytestmark = pytest.mark.gpu
SUPPORTS = [(-300., 301., 1.), (-50., 51., 1.), (-10., 11., 1.), (-2., 3., 1.)]


def _case(seed):
    r = np.random.default_rng(4000 + seed)
    kind = ["ez_atari", "mz_atari", "mz_board", "ez_board"][int(r.integers(0, 4))]
    sup = SUPPORTS[int(r.integers(0, len(SUPPORTS)))]
    kw = dict(num_res_blocks=int(r.integers(1, 4)), reward_support_range=sup, value_support_range=sup)
    if kind.endswith("atari"):
        hw = int(r.choice([96, 64]))
        kw.update(observation_shape=(int(r.choice([4, 1, 3])), hw, hw), action_space_size=int(r.integers(2, 19)), downsample=True)
    else:
        gh, gw = [(3, 3), (6, 6), (6, 7), (9, 9), (8, 8), (4, 4)][int(r.integers(0, 6))]
        kw.update(observation_shape=(int(r.integers(1, 18)), gh, gw), action_space_size=int(r.integers(2, gh * gw + 2)), downsample=False)
        if r.random() < 0.4 and (gh, gw) != (8, 8):
            kw["num_channels"] = int(r.choice([32, 16]))
    if r.random() < 0.25:
        kw["discrete_action_encoding_type"] = "not_one_hot"
    return dict(family="ez" if kind.startswith("ez") else "mz", kw=kw, B=int(r.integers(1, 41)), seed=600 + seed)


def _oracle_outputs(case, model, forced=None):
    """the golden file's arrays, computed by the torch restatement (tests/golden/make_golden_nn.py does the same on the reference modules).
    ``forced``: the arrays of an earlier (fp32) run -- every step then consumes THAT run's states, cast to this model's dtype: the binary64
    evaluation of exactly the inputs the checker feeds the engine"""
    from oracle import torch_models as tm
    kw = case["kw"]
    dt = next(model.parameters()).dtype
    ist = tm.InverseScalarTransform(kw["value_support_range"])
    rist = tm.InverseScalarTransform(kw["reward_support_range"])
    ist.value_support, rist.value_support = ist.value_support.to(dt), rist.value_support.to(dt)
    obs, actions = nn_cases.inputs(case)
    lstm = nn_cases.has_lstm(case["family"])
    out = {}
    with torch.no_grad():
        r = model.initial_inference(torch.from_numpy(obs).to(dt))
        out["init_latent"], out["init_value_logits"] = r.latent_state.numpy(), r.value.numpy()
        out["init_value"], out["init_policy"] = ist(r.value.clone()).reshape(-1).numpy(), r.policy_logits.numpy()
        lat, hc = r.latent_state, (r.reward_hidden_state if lstm else None)
        for s in range(nn_cases.STEPS):
            a = torch.from_numpy(actions[s])
            if forced is not None:
                lat = torch.from_numpy(forced["s%d_in_latent" % s]).to(dt)
                if lstm:
                    hc = (torch.from_numpy(forced["s%d_in_h" % s]).to(dt)[None], torch.from_numpy(forced["s%d_in_c" % s]).to(dt)[None])
            out["s%d_in_latent" % s] = lat.numpy()
            if lstm:
                out["s%d_in_h" % s], out["s%d_in_c" % s] = hc[0][0].numpy(), hc[1][0].numpy()
                r = model.recurrent_inference(lat, hc, a)
                out["s%d_h" % s], out["s%d_c" % s] = r.reward_hidden_state[0][0].numpy(), r.reward_hidden_state[1][0].numpy()
                rew_logits, hc = r.value_prefix, r.reward_hidden_state
            else:
                r = model.recurrent_inference(lat, a)
                rew_logits = r.reward
            out["s%d_latent" % s], out["s%d_reward_logits" % s] = r.latent_state.numpy(), rew_logits.numpy()
            out["s%d_reward" % s] = rist(rew_logits.clone()).reshape(-1).numpy()
            out["s%d_value_logits" % s], out["s%d_value" % s] = r.value.numpy(), ist(r.value.clone()).reshape(-1).numpy()
            out["s%d_policy" % s] = r.policy_logits.numpy()
            lat = r.latent_state
    return out


def _fp32_cost(g32, g64):
    """per tensor class: what fp32 arithmetic itself loses on this network -- torch fp32 against binary64 on the same inputs"""
    def rel(k):
        a, b = np.asarray(g32[k], np.float64), np.asarray(g64[k], np.float64)
        return float(np.max(np.abs(a - b) / (1.0 + np.abs(b))))
    cls = dict(latent=[0.0], policy=[0.0], scalar=[0.0], logits=[0.0], hc=[0.0])
    for k in g32:
        if "_in_" in k:
            continue
        for suffix, c in (("_latent", "latent"), ("_policy", "policy"), ("_logits", "logits"), ("_value", "scalar"), ("_reward", "scalar"),
                          ("_h", "hc"), ("_c", "hc")):
            if k.endswith(suffix):
                cls[c].append(rel(k))
                break
    return {k: max(v) for k, v in cls.items()}


@pytest.mark.parametrize("seed", range(_OFF, _OFF + 24))
def test_random_model_configuration_matches_the_torch_restatement(seed):
    from oracle import torch_models as tm
    from lightzero_amd import _lib as L
    case = _case(seed)
    model = tm.synthetic_init(nn_cases.oracle_class(tm, case["family"])(**case["kw"]), seed=case["seed"]).eval()
    g32 = _oracle_outputs(case, model)
    cost = _fp32_cost(g32, _oracle_outputs(case, copy.deepcopy(model).double(), forced=g32))
    bounds = {k: max(parity_record.BOUNDS[k], 3.0 * cost[k]) for k in cost}
    print("fp32 cost of this network (torch fp32 vs binary64):", cost)
    try:
        check_case("fuzz%02d" % seed, case, g32, model.state_dict(), record="fuzz/", bounds=bounds)
    except (L.LzError, ValueError, NotImplementedError) as e:
        assert len(str(e)) > 20, repr(e)
        pytest.skip("refused by the engine: %s" % e)


def _sez_case(seed):
    r = np.random.default_rng(4500 + seed)
    hw, hid, A = int(r.choice([64, 96])), int(r.choice([32, 64, 128, 256])), int(r.integers(2, 19))
    sup = SUPPORTS[int(r.integers(0, 2))]
    kw = dict(observation_shape=(4, hw, hw), action_space_size=A, num_of_sampled_actions=int(r.integers(1, A + 1)), downsample=True,
              continuous_action_space=False, norm_type='BN', num_res_blocks=int(r.integers(1, 4)), reward_support_range=sup, value_support_range=sup,
              reward_head_hidden_channels=[hid], value_head_hidden_channels=[hid], policy_head_hidden_channels=[hid])
    if r.random() < 0.5:
        kw["activation"] = "relu"
    return dict(family="sez", kw=kw, B=int(r.integers(1, 41)), seed=700 + seed)


@pytest.mark.parametrize("seed", range(_OFF, _OFF + 8))
def test_random_conv_sampled_model_matches_the_torch_restatement(seed):
    """the convolutional Sampled EfficientZero (round 4: GELU(tanh) / ReLU dynamics, GELU prediction network, head widths 32..256)"""
    from oracle import torch_models as tm
    case = _sez_case(seed)
    model = tm.synthetic_init(nn_cases.oracle_class(tm, "sez")(**case["kw"]), seed=case["seed"]).eval()
    g32 = _oracle_outputs(case, model)
    cost = _fp32_cost(g32, _oracle_outputs(case, copy.deepcopy(model).double(), forced=g32))
    bounds = {k: max(parity_record.BOUNDS[k], 3.0 * cost[k]) for k in cost}
    print("fp32 cost of this network (torch fp32 vs binary64):", cost)
    check_case("fuzz_sez%02d" % seed, case, g32, model.state_dict(), record="fuzz/", bounds=bounds)


# ---- the three networks the round-4 / round-5 sweeps left above their bound, as named cases of the committed suite.
# Round 6 found the layer by switching ONE kernel (tools/r06_s3g.sh, profiles/r06_named_networks.json): all three ran an fp32-matrix chain --
# 1447: three residual blocks on a 6x7 board through k_chain<7,6> (one 576-term fp32 accumulation chain per output); sez600 / sez14: 8x8 latents
# through the Winograd chain k_chain_w<8,8> (its input / output transforms add roundings of their own) -- and all three are inside their bound
# on the split-bf16 chain k_chain_s3g (exact plane products, four / two partial sums per output), with LZ_CHAIN_NO_SPLIT=1 reproducing the
# old values to the last digit.  The ceilings below are the round-6 measurements plus 15 %: a kernel change that loses accuracy on exactly
# these networks fails here instead of living in an offset nobody runs.
NAMED = [
    # (name, kind, seed, {tensor class: ceiling on |device - torch fp32| / (1 + |x|)})
    ("fuzz1447", "conv", 1447, dict(policy=5.4e-6, logits=3.3e-6)),     # round 5: policy 1.026e-5 against 1e-5
    ("fuzz_sez600", "sez", 600, dict(policy=2.71e-5, latent=2.11e-5)),  # round 5: policy 3.28e-5 against 3.12e-5
    ("fuzz_sez14", "sez", 14, dict(policy=8.4e-6, hc=1.25e-5)),         # round 5: scalar 9.35e-4 against 8.69e-4 (post-h^-1: held to the sweep's bound)
    # Round 6, offset 56 of the final sweep (profiles/r06_parity_sweeps.json: 1 of 1743 networks): 4x64x64 -> 8x8 latent, THREE residual blocks,
    # ReLU dynamics / GELU prediction, 64-wide heads, B = 29 -- a network on which torch's own fp32 evaluation is 1.9e-5 (latent) / 1.5e-5 (logits)
    # from a binary64 evaluation of the same inputs.  |device - torch| is 1.525e-5 on the policy logits against the sweep's 1.415e-5 (= 3 x torch's
    # own 4.7e-6) and 6.6e-4 against 5.1e-4 on the post-h^-1 scalars.  Found by switching kernels (tools/r06_sez63.py, profiles/r06_sez63.json): only
    # the recurrent CHAIN moves the numbers, and BOTH of its arithmetic forms are outside (split-bf16 k_chain_s3g: policy 1.53e-5; the fp32 chain
    # k_chain_w<8,8>: 1.84e-5) -- the tower, LSTM and head switches leave every digit as it is.  Against the binary64 values the device is INSIDE the
    # same 3 x rule in every class (policy 1.20e-5 <= 1.41e-5, latent 2.49e-5 <= 5.64e-5, logits 2.17e-5 <= 4.64e-5, h / c 1.38e-5 <= 7.45e-5): device
    # and torch err in different directions, and |device - torch| adds the two.  Held here to that anchor and to ceilings on the torch distances.
    ("fuzz_sez63", "sez", 63, dict(policy=1.76e-5, latent=3.53e-5, logits=2.96e-5, hc=3.34e-5, scalar=7.61e-4)),
]
ANCHORED_TO_BINARY64 = {"fuzz_sez63"}   # networks above the sweep's |device - torch| bound: asserted against the binary64 evaluation instead


@pytest.mark.parametrize("name,kind,seed,ceil", NAMED, ids=[n[0] for n in NAMED])
def test_named_network_stays_inside_its_bound(name, kind, seed, ceil):
    import os
    from oracle import torch_models as tm
    case = _case(seed) if kind == "conv" else _sez_case(seed)
    model = tm.synthetic_init(nn_cases.oracle_class(tm, case["family"])(**case["kw"]), seed=case["seed"]).eval()
    g32 = _oracle_outputs(case, model)
    g64 = _oracle_outputs(case, copy.deepcopy(model).double(), forced=g32)
    cost = _fp32_cost(g32, g64)
    bounds = {k: max(parity_record.BOUNDS[k], 3.0 * cost[k]) for k in cost}
    held = dict(bounds)
    if name in ANCHORED_TO_BINARY64:   # |device - torch| is held to the ceilings, the sweep's 3 x rule to |device - binary64| below
        held = {k: max(bounds[k], ceil.get(k, bounds[k])) for k in bounds}
    worst = check_case(name, case, g32, model.state_dict(), record="named/", bounds=held, g64=g64)
    parity_record.record("named/" + name, {}, extra=dict(torch_fp32_vs_binary64=cost))
    if name in ANCHORED_TO_BINARY64:
        far = {k: (v, bounds[k]) for k, v in worst["vs64"].items() if not v <= bounds[k]}
        assert not far, "%s: further from the binary64 evaluation than 3 x torch fp32 is (measured, bound): %s" % (name, far)
    if not any(os.environ.get(k) for k in ("LZ_CHAIN_NO_SPLIT", "LZ_CHAIN_DIRECT")):   # (the old kernels are allowed their old values)
        bad = {k: (worst[k], c) for k, c in ceil.items() if not worst[k] <= c}
        assert not bad, "%s: above its round-6 ceiling (measured, ceiling): %s" % (name, bad)
