"""HIP network kernels (through the C ABI) against the committed outputs of the REFERENCE's own model modules
(tests/golden/nn_*.npz, tests/golden/make_golden_nn.py): initial_inference on the seeded observations, then every recurrent
step TEACHER-FORCED -- the reference's own (latent, h, c) of the previous step is written into a pool slot
(lz_roots_write_latent / _hidden) and lz_recurrent_inference runs on it, so each step is compared on identical inputs.

Tolerances (fp32 everywhere, only summation order differs; tests/parity_record.py holds the table and keeps
the MEASURED worst case of every run as data -> profiles/rNN_parity.json):
  latent / LSTM state / policy logits / support-wide logits: |d| <= 1e-5 (1 + |x|)   (north_star's bound)
  value / value-prefix / reward scalars after h^-1:           |d| <= 3e-4 (1 + |x|)   (the reference's own fp32 formula quantises
                                                              its output in steps of ~1.3e-4 (1 + |x|))"""
import os
import sys

import numpy as np
import pytest

import nn_cases
import parity_record

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a - b) / (1.0 + np.abs(b))))


def _roots_for(case, model, B):
    fam, kw = case["family"], case["kw"]
    if fam == "sez_mlp":
        from lightzero_amd.mcts.ctree.ctree_sampled_efficientzero import ezs_tree
        K = kw["num_of_sampled_actions"]
        return ezs_tree.Roots(B, [[-1] * K] * B, kw["action_space_size"], K, kw.get("continuous_action_space", True),
                              max_simulations=4, engine=model.engine)
    if fam == "sez":   # the convolutional Sampled EfficientZero (discrete actions): the sampled tree over the model's action space
        from lightzero_amd.mcts.ctree.ctree_sampled_efficientzero import ezs_tree
        K, A = kw["num_of_sampled_actions"], kw["action_space_size"]
        return ezs_tree.Roots(B, [list(range(A))] * B, A, K, False, max_simulations=4, engine=model.engine)
    if fam in ("ez", "ez_mlp"):
        from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree as tree
    else:
        from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree as tree
    A = kw["action_space_size"]
    return tree.Roots(B, [list(range(A))] * B, action_space_size=A, max_simulations=4, engine=model.engine)


@pytest.mark.parametrize("name", sorted(nn_cases.CASES))
def test_hip_network_matches_reference_module_outputs(name):
    from make_golden_nn import weights_digest
    from oracle import torch_models as tm
    case = nn_cases.CASES[name]
    fam, kw = case["family"], case["kw"]
    g = np.load(os.path.join(GOLD, "nn_%s.npz" % name))
    sd = tm.synthetic_init(nn_cases.oracle_class(tm, fam)(**kw), seed=case["seed"]).state_dict()  # the seeded weight recipe only
    assert weights_digest(sd) == bytes(g["weights_sha256"]).decode(), "seeded weights differ from the golden's"
    check_case(name, case, g, sd)


def check_case(name, case, g, sd, record="golden/", bounds=None, g64=None, engine_kw=None, keep=None):
    """engine model with the weights ``sd`` on the seeded inputs of ``case`` against the arrays ``g`` (the golden file's layout).
    ``g64``: the same arrays from a binary64 evaluation of the same network on the same teacher-forced inputs -- the device's distance
    from THEM is recorded next to its distance from the fp32 reference (entry field "device_vs_binary64") and returned under "vs64"."""
    from lightzero_amd import _lib as L
    lib = L.lib()
    fam, kw, B = case["family"], case["kw"], case["B"]
    ekw = dict(kw)
    if fam in ("mz_mlp", "ez_mlp") and "norm_type" not in ekw:
        ekw["norm_type"] = "BN"
    ekw.update(engine_kw or {})   # arguments of the engine model only (fast_mode, fp32_matrix)
    model = nn_cases.engine_class(fam)(**ekw).load_state_dict(sd)
    roots = _roots_for(case, model, B)
    obs, actions = nn_cases.inputs(case)
    if fam not in ("sez_mlp", "sez"):   # (the sampled roots create their device handle in the constructor)
        roots._bind_engine(model.engine)
        roots._ensure(kw["action_space_size"])
    L.check(lib.lz_roots_enable_trace(roots._h, 1))   # the heads also write their support-wide logits
    conv = True   # the vector-observation models expose their support-wide logits the same way
    out = model.initial_inference(obs, roots)
    lat = np.zeros(g["init_latent"].shape, np.float32)
    L.check(lib.lz_roots_read_latent(roots._h, 0, lat.reshape(-1)))
    if keep is not None:
        keep["init_latent"] = lat.copy()
    worst = dict(latent=_rel(lat, g["init_latent"]), policy=_rel(out.policy_logits, g["init_policy"]),
                 scalar=_rel(out.value, g["init_value"]), logits=0.0, hc=0.0)
    w64 = dict(latent=0.0, policy=0.0, logits=0.0, hc=0.0)
    def up64(cls, dev, key):
        if g64 is not None:
            w64[cls] = max(w64[cls], _rel(np.asarray(dev, np.float64), np.asarray(g64[key], np.float64)))
    up64("latent", lat, "init_latent"); up64("policy", out.policy_logits, "init_policy")
    SUP, RSUP = g["init_value_logits"].shape[1], g["s0_reward_logits"].shape[1]
    if conv:
        vl = np.zeros((B, SUP), np.float32)
        L.check(lib.lz_roots_read_debug_logits(roots._h, 0, vl.reshape(-1)))
        worst["logits"] = _rel(vl, g["init_value_logits"])
    zeros = np.zeros(B, np.int32)
    lstm = nn_cases.has_lstm(fam)
    PA = g["init_policy"].shape[1]
    for s in range(nn_cases.STEPS):
        L.check(lib.lz_roots_write_latent(roots._h, 0, np.ascontiguousarray(g["s%d_in_latent" % s]).reshape(-1)))
        if lstm:
            L.check(lib.lz_roots_write_hidden(roots._h, 0, np.ascontiguousarray(g["s%d_in_h" % s]).reshape(-1),
                                              np.ascontiguousarray(g["s%d_in_c" % s]).reshape(-1)))
        a = actions[s]
        if a.dtype == np.float32:
            a = np.ascontiguousarray(a)
            L.check(lib.lz_recurrent_inference(roots._h, zeros, None, a.ctypes.data, None, 0, 1))
        else:
            a = np.ascontiguousarray(a, np.int32)
            af = np.ascontiguousarray(a, np.float32)  # sampled roots with a discrete action space carry the index as a float
            if fam in ("sez_mlp", "sez"):
                L.check(lib.lz_recurrent_inference(roots._h, zeros, None, af.ctypes.data, None, 0, 1))
            else:
                L.check(lib.lz_recurrent_inference(roots._h, zeros, a.ctypes.data, None, None, 0, 1))
        L.check(lib.lz_roots_read_latent(roots._h, 1, lat.reshape(-1)))
        rew = np.zeros(B, np.float32); val = np.zeros(B, np.float32); pol = np.zeros((B, PA), np.float32)
        L.check(lib.lz_roots_read_sim_outputs(roots._h, 1, rew, val, pol.reshape(-1)))
        worst["latent"] = max(worst["latent"], _rel(lat, g["s%d_latent" % s]))
        worst["policy"] = max(worst["policy"], _rel(pol, g["s%d_policy" % s]))
        up64("latent", lat, "s%d_latent" % s); up64("policy", pol, "s%d_policy" % s)
        worst["scalar"] = max(worst["scalar"], _rel(rew, g["s%d_reward" % s]), _rel(val, g["s%d_value" % s]))
        if lstm:
            H = g["s%d_h" % s].shape[1]
            hh = np.zeros((B, H), np.float32); cc = np.zeros((B, H), np.float32)
            L.check(lib.lz_roots_read_hidden(roots._h, 1, hh.reshape(-1), cc.reshape(-1)))
            worst["hc"] = max(worst["hc"], _rel(hh, g["s%d_h" % s]), _rel(cc, g["s%d_c" % s]))
            up64("hc", hh, "s%d_h" % s); up64("hc", cc, "s%d_c" % s)
        if conv:
            vl = np.zeros((B, SUP), np.float32); rl = np.zeros((B, RSUP), np.float32)
            L.check(lib.lz_roots_read_debug_logits(roots._h, 0, vl.reshape(-1)))
            L.check(lib.lz_roots_read_debug_logits(roots._h, 1, rl.reshape(-1)))
            worst["logits"] = max(worst["logits"], _rel(vl, g["s%d_value_logits" % s]), _rel(rl, g["s%d_reward_logits" % s]))
            up64("logits", vl, "s%d_value_logits" % s); up64("logits", rl, "s%d_reward_logits" % s)
    print(name, "worst relative differences:", worst, ("device vs binary64: %s" % w64) if g64 is not None else "")
    extra = dict(batch=int(B), steps=int(nn_cases.STEPS))
    if g64 is not None:
        extra["device_vs_binary64"] = w64
    parity_record.check(record + name, worst, extra=extra, bounds=bounds)
    if g64 is not None:
        worst = dict(worst, vs64=w64)
    return worst


@pytest.mark.parametrize("name", ["ez_atari96", "ez_atari64", "mz_go9"])
def test_fp32_matrix_arithmetic_is_a_model_argument(name):
    """lz_model_cfg.precision = 2 (EfficientZeroModel(fp32_matrix=True)): parity mode on the fp32 matrix instructions -- the per-model form of
    LZ_CHAIN_NO_SPLIT=1 LZ_CONV_NO_SPLIT=1 (the choice was a process-wide environment switch).  The same goldens at the same bounds
    as the default (split-bf16) arithmetic, in the same process, and other bits than the default path's -- the other kernels did run."""
    from oracle import torch_models as tm
    case = nn_cases.CASES[name]
    g = np.load(os.path.join(GOLD, "nn_%s.npz" % name))
    sd = tm.synthetic_init(nn_cases.oracle_class(tm, case["family"])(**case["kw"]), seed=case["seed"]).state_dict()
    a, b = {}, {}
    check_case(name, case, g, sd, record="golden_fp32_matrix/", engine_kw=dict(fp32_matrix=True), keep=a)
    check_case(name, case, g, sd, keep=b)
    assert a["init_latent"].shape == b["init_latent"].shape
    # (under the process-wide switches that take the split kernels away from the DEFAULT model too -- tests/test_kernel_variants_gpu.py runs this
    # file with each of them -- the two models run the same kernels)
    if not any(os.environ.get(k) for k in ("LZ_CHAIN_NO_SPLIT", "LZ_CONV_NO_SPLIT", "LZ_CHAIN_DIRECT", "LZ_CHAIN_W4", "LZ_CONV_DIRECT")):
        assert not np.array_equal(a["init_latent"], b["init_latent"])
