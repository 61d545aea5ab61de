"""GPU: every configuration file the reference ships for the four families on the hot path (tests/golden/zoo_model_configs.json, made by
tests/golden/make_zoo_model_configs.py from zoo/**/config/*.py: the model keywords each file sets, completed with the REFERENCE class's
defaults) is built on the engine with exactly those keywords.

* accepted: seeded weights go in by the reference's parameter names and the engine's initial inference + three teacher-forced recurrent
  inferences are held to the torch restatement of the same keywords (oracle/torch_models.py, pinned bit-equal to the reference modules by
  tests/test_torch_models_vs_reference.py) at the bounds of the randomised sweep: 1e-5 (1 + |x|) before h^-1, or 3 x what torch's own fp32
  evaluation loses against binary64 on that network where that is more;
* refused: the engine must say why (NotImplementedError / LzError with a message), and a configuration the restatement cannot express
  (a keyword it has no counterpart for, set to something that changes the graph) must be among the refused.

The outcome per file is written to gpurun_out/zoo_configs.json (committed copy: profiles/rNN_zoo_configs.json)."""
import copy
import inspect
import json
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

import nn_cases
import parity_record

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
CONFIGS = json.load(open(os.path.join(HERE, "golden", "zoo_model_configs.json")))["configs"]

# keywords of the reference constructors that do not shape the inference graph (projection / prediction heads of the self-supervised
# loss, initialisation switches, bookkeeping the env wrappers read)
TRAINING_ONLY = {"self_supervised_learning_loss", "last_linear_layer_init_zero", "pred_hid", "pred_out", "proj_hid", "proj_out",
                 "image_channel", "frame_stack_num", "gray_scale", "model_type", "analysis_sim_norm"}
# keywords the restatement has no argument for because it is written for ONE value of them: that value
NEUTRAL = {"activation": ("ReLU", "GELU(tanh)"), "norm_type": ("BN",), "categorical_distribution": (True,), "state_norm": (False,),
           "use_sim_norm": (False,), "bound_type": (None,), "fixed_sigma_value": (0.3,), "sigma_type": ("conditioned",),
           "continuous_action_space": (False,), "downsample": (True,), "discrete_action_encoding_type": ("one_hot",),
           "res_connection_in_dynamics": (True,)}


def _family(e):
    return {"conv": e["family"], "mlp": e["family"] + "_mlp"}.get(e["model_type"])


def _accepted_keys(cls):
    """named parameters of cls.__init__, following **kw pass-through up the MRO"""
    keys = set()
    for c in cls.__mro__:
        init = c.__dict__.get("__init__")
        if init is None:
            continue
        ps = inspect.signature(init).parameters
        keys |= {k for k, p in ps.items() if p.kind in (p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY) and k != "self"}
        if not any(p.kind == p.VAR_KEYWORD for p in ps.values()):
            break
    return keys


def _module(v):
    if v == "ReLU":
        return nn.ReLU(inplace=True)
    if isinstance(v, str) and v.startswith("GELU"):
        return nn.GELU(approximate=v[5:-1] or "none")
    return v


def _oracle_kwargs(fam, ref_kw, tm):
    """-> (kwargs for the restatement, [keywords it cannot express])"""
    keys = _accepted_keys(nn_cases.oracle_class(tm, fam))
    kw, lost = {}, []
    for k, v in ref_kw.items():
        if k in keys:
            kw[k] = _module(v) if k == "activation" else (tuple(v) if isinstance(v, list) and k.endswith("_range") else v)
        elif k in TRAINING_ONLY or (k in NEUTRAL and v in NEUTRAL[k]):
            continue
        elif k in ("value_support_range", "reward_support_range") and "support_range" in keys:
            continue    # handled below
        else:
            lost.append("%s=%r" % (k, v))
    if "support_range" in keys and "value_support_range" not in keys and "value_support_range" in ref_kw:
        if ref_kw.get("reward_support_range", ref_kw["value_support_range"]) != ref_kw["value_support_range"]:
            lost.append("reward_support_range != value_support_range")
        kw["support_range"] = tuple(ref_kw["value_support_range"])
    if isinstance(kw.get("observation_shape"), list):
        kw["observation_shape"] = tuple(kw["observation_shape"])
    return kw, lost


def _arrays(case, model, tm, forced=None):
    """the golden files' arrays (tests/golden/make_golden_nn.py) from the torch restatement; ``forced``: an earlier run's states (binary64
    evaluation of exactly the inputs the checker feeds the engine)"""
    kw, fam = case["kw"], case["family"]
    dt = next(model.parameters()).dtype
    support = kw.get("value_support_range", kw.get("support_range", (-300., 301., 1.)))
    cat = bool(kw.get("categorical_distribution", True))
    ist = tm.InverseScalarTransform(support, cat)
    rist = tm.InverseScalarTransform(kw.get("reward_support_range") or support, cat)
    ist.value_support, rist.value_support = ist.value_support.to(dt), rist.value_support.to(dt)
    obs, actions = nn_cases.inputs(case)
    lstm = nn_cases.has_lstm(fam)
    out = {}
    with torch.no_grad():
        r = model.initial_inference(torch.from_numpy(obs).to(dt))
        out["init_latent"], out["init_value_logits"] = r.latent_state.numpy(), r.value.numpy()
        out["init_value"], out["init_policy"] = ist(r.value.clone()).reshape(-1).numpy(), r.policy_logits.numpy()
        lat, hc = r.latent_state, (r.reward_hidden_state if lstm else None)
        for s in range(nn_cases.STEPS):
            a = torch.from_numpy(actions[s])
            if a.dtype.is_floating_point:
                a = a.to(dt)
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


def _write(path_rel, outcome):
    import fcntl
    out = os.path.join(os.path.dirname(HERE), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "zoo_configs.json")
    with open(path + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            data = json.load(open(path))
        except Exception:
            data = {}
        data[path_rel] = outcome
        json.dump(data, open(path, "w"), indent=1, sort_keys=True)


@pytest.mark.parametrize("path_rel", sorted(CONFIGS))
def test_shipped_configuration_is_served_or_refused_with_a_reason(path_rel):
    from oracle import torch_models as tm
    from lightzero_amd import _lib as L
    from test_nn_golden_gpu import check_case
    from test_nn_fuzz_gpu import _fp32_cost
    e = CONFIGS[path_rel]
    fam = _family(e)
    base = dict(policy_type=e["policy_type"], model_type=e["model_type"], reference_class=e["reference_class"])
    if fam is None:
        # another model class of the reference (model_type 'conv_context' -> MuZeroContextModel): not one of the engine's
        _write(path_rel, dict(base, served=False, reason="model_type %r builds a reference class outside the MuZero / EfficientZero / Sampled "
                                                        "EfficientZero conv | mlp models" % e["model_type"]))
        return
    ref_kw = dict(e["model"])
    okw, lost = _oracle_kwargs(fam, ref_kw, tm)
    ekw = {k: (_module(v) if k == "activation" else v) for k, v in ref_kw.items()}
    ref_model, build_err = None, None
    if not lost:
        try:
            ref_model = tm.synthetic_init(nn_cases.oracle_class(tm, fam)(**okw), seed=77).eval()
        except (AssertionError, NotImplementedError, ValueError, RuntimeError, TypeError) as ex:
            build_err = "%s: %s" % (type(ex).__name__, ex)
    try:
        model = nn_cases.engine_class(fam)(**ekw)
        if ref_model is not None:
            model.load_state_dict(ref_model.state_dict())
    except (L.LzError, NotImplementedError, ValueError) as ex:
        assert len(str(ex)) > 20, repr(ex)
        _write(path_rel, dict(base, served=False, reason=str(ex)[:400]))
        return
    # accepted: then the restatement must be able to say what the right answer is
    assert not lost, "the engine accepted keywords the restatement cannot express: %s" % lost
    assert ref_model is not None, "the engine accepted a configuration the restatement refuses: %s" % build_err
    del model
    case = dict(family=fam, kw=okw, B=4, seed=77)
    g32 = _arrays(case, ref_model, tm)
    try:
        cost = _fp32_cost(g32, _arrays(case, copy.deepcopy(ref_model).double(), tm, forced=g32))
        bounds = {k: max(parity_record.BOUNDS[k], 3.0 * cost[k]) for k in cost}
    except RuntimeError:    # the vector-observation restatements cast their input to fp32 (as the reference modules do): no binary64 twin
        bounds = {k: parity_record.BOUNDS[k] for k in ("latent", "policy", "scalar", "logits", "hc")}
    name = os.path.basename(path_rel)[:-3]
    worst = check_case(name, case, g32, ref_model.state_dict(), record="zoo/", bounds=bounds, engine_kw={k: v for k, v in ekw.items() if k not in okw})
    outcome = dict(base, served=True, worst={k: float(v) for k, v in worst.items() if not isinstance(v, dict)},
                   bounds={k: float(v) for k, v in bounds.items()})
    outcome["search"] = _search_stage(e, fam, okw, ekw, ref_model)
    outcome["policy_surface"] = _policy_stage(e, fam, okw, ekw, ref_model)
    _write(path_rel, outcome)


def _policy_stage(e, fam, okw, ekw, ref_model):
    """the reference-named policy class built from the FILE'S OWN policy dictionary (num_simulations, eps, env_type, discount, root noise ...):
    _forward_collect and _forward_eval on the file's collector batch return the reference's per-env dictionary -- a legal action, visit
    counts over the legal (or sampled) actions that sum to num_simulations, the value / entropy / logits fields"""
    from lightzero_amd.policy.efficientzero import EfficientZeroPolicy
    from lightzero_amd.policy.muzero import MuZeroPolicy
    from lightzero_amd.policy.sampled_efficientzero import SampledEfficientZeroPolicy
    from lightzero_amd.policy.gumbel_muzero import GumbelMuZeroPolicy
    cls = {"efficientzero": EfficientZeroPolicy, "muzero": MuZeroPolicy, "sampled_efficientzero": SampledEfficientZeroPolicy,
           "gumbel_muzero": GumbelMuZeroPolicy}[e["policy_type"]]
    B, S = int(e["collector_env_num"] or 8), int(e["num_simulations"] or 50)
    A = int(okw["action_space_size"])
    board = e["env_type"] == "board_games"
    sampled = e["policy_type"] == "sampled_efficientzero"
    cont = sampled and bool(okw.get("continuous_action_space", False))
    K = int(okw.get("num_of_sampled_actions", 0))
    model = nn_cases.engine_class(fam)(**ekw).load_state_dict(ref_model.state_dict())
    pcfg = dict(e.get("policy") or {})
    pcfg["model"] = dict(e["model"])
    policy = cls(pcfg, model)
    rng = np.random.default_rng(11)
    shape = okw["observation_shape"]
    g = torch.Generator().manual_seed(12)
    obs = (torch.randn(B, shape, generator=g) if isinstance(shape, int) else torch.rand(B, *shape, generator=g)).cuda().contiguous()
    if cont:
        mask = [None] * B
        legal_n = [K] * B
    else:
        m2 = np.ones((B, A), np.int8)
        if board:
            m2 = (rng.random((B, A)) < 0.7).astype(np.int8)
            m2[np.arange(B), rng.integers(0, A, size=B)] = 1
        mask = [m2[i] for i in range(B)]
        legal_n = m2.sum(1).tolist()
    to_play = rng.integers(1, 3, size=B).tolist() if board and e["policy_type"] != "gumbel_muzero" else [-1] * B
    ids = np.arange(100, 100 + B)
    for which in ("collect", "eval"):
        if which == "collect":
            out = policy._forward_collect(obs, action_mask=mask, temperature=1.0, to_play=to_play, epsilon=0.25, ready_env_id=ids)
        else:
            out = policy._forward_eval(obs, action_mask=mask, to_play=to_play, ready_env_id=ids)
        assert sorted(out) == sorted(ids.tolist()), "output not keyed by ready_env_id"
        pure = which == "collect" and bool(pcfg.get("collect_with_pure_policy", False))
        for i, env_id in enumerate(ids.tolist()):
            o = out[env_id]
            if pure:   # efficientzero.py:644-656: no search in the collect forward -- an action sampled from the policy, four fields
                assert sorted(o) == ["action", "predicted_policy_logits", "predicted_value", "searched_value"], sorted(o)
                assert mask[i][int(o["action"])] == 1
                continue
            for key in ("action", "visit_count_distributions", "visit_count_distribution_entropy", "searched_value", "predicted_value", "predicted_policy_logits"):
                assert key in o, (which, key, sorted(o))
            dist = list(o["visit_count_distributions"])
            assert sum(dist) == S, (which, "visit counts sum to %d, not num_simulations %d" % (sum(dist), S))
            if sampled:
                assert len(dist) == K and "root_sampled_actions" in o
            else:
                assert len(dist) == legal_n[i]
                assert mask[i][int(o["action"])] == 1, (which, "illegal action")
            assert np.isfinite(float(np.asarray(o["searched_value"]).reshape(-1)[0]))
    return "%s(cfg.policy, model): _forward_collect + _forward_eval on %d envs, %d simulations" % (cls.__name__, B, S)


def _search_stage(e, fam, okw, ekw, ref_model):
    """the file's own collector batch and simulation count through the fused search, replayed exactly through the oracle trees (and the
    reference's compiled ctree where the box has it) with the device's own network outputs: identical visit counts, bit-equal values /
    min-max statistics, identical per-simulation records"""
    from test_exact_replay_gpu import _search_and_replay
    B, S = int(e["collector_env_num"] or 8), int(e["num_simulations"] or 50)
    A = int(okw["action_space_size"])
    board = e["env_type"] == "board_games"
    discount = float(e["discount_factor"])
    rng = np.random.default_rng(5)
    model = nn_cases.engine_class(fam)(**ekw).load_state_dict(ref_model.state_dict())
    shape = okw["observation_shape"]
    g = torch.Generator().manual_seed(9)
    obs = (torch.randn(B, shape, generator=g) if isinstance(shape, int) else torch.rand(B, *shape, generator=g)).cuda().contiguous()
    what = "%d roots x %d simulations, discount %g%s" % (B, S, discount, ", two players, ragged legal actions" if board else "")
    if e["policy_type"] == "gumbel_muzero":
        # (the reference's Gumbel driver has no two-player mode: to_play stays -1; ragged legal lists on the boards)
        from lightzero_amd.mcts.tree_search.mcts_ctree import GumbelMuZeroMCTSCtree
        from test_exact_replay_families_gpu import _gumbel_search_and_replay
        m = int(e.get("max_num_considered_actions") or min(A, 16))
        mcts = GumbelMuZeroMCTSCtree(dict(num_simulations=S, discount_factor=discount, max_num_considered_actions=m, value_delta_max=0.01,
                                          root_noise_weight=0.25))
        legal = []
        for _ in range(B):
            k = rng.random(A) < (0.7 if board else 2.0)
            k[rng.integers(0, A)] = True
            legal.append(np.nonzero(k)[0].tolist())
        noises = [rng.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal]
        _gumbel_search_and_replay(model, mcts, B, A, S, m, legal, noises, obs, discount=discount)
        return "exact replay (Gumbel, m = %d): %s" % (m, what.replace(", two players", ""))
    if fam in ("mz", "ez", "mz_mlp", "ez_mlp"):
        if fam.startswith("ez"):
            from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree as tree
        else:
            from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree as tree
        if board:
            legal = []
            for _ in range(B):
                m = rng.random(A) < 0.7
                m[rng.integers(0, A)] = True
                legal.append(np.nonzero(m)[0].tolist())
            to_play = rng.integers(1, 3, size=B).tolist()
        else:
            legal, to_play = [list(range(A))] * B, [-1] * B
        noises = [rng.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal]
        roots = tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
        roots.set_tiebreak(0)
        _search_and_replay(fam[:2], model, roots, obs, legal, to_play, noises, S, discount, trace=True)
        return "exact replay: " + what
    # Sampled EfficientZero: the K actions of every node are drawn on the device inside the captured graph, read back and injected into the oracle tree
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.tree_search.mcts_ctree import SampledEfficientZeroMCTSCtree
    from test_exact_replay_families_gpu import _sampled_replay
    K, cont = int(okw["num_of_sampled_actions"]), bool(okw.get("continuous_action_space", False))
    cfg = dict(num_simulations=S, pb_c_base=19652, pb_c_init=1.25, discount_factor=0.997, value_delta_max=0.01, lstm_horizon_len=5,
               root_noise_weight=0.25, model=dict(action_space_size=A, num_of_sampled_actions=K, continuous_action_space=cont))
    mcts = SampledEfficientZeroMCTSCtree(cfg)
    roots = mcts.roots(B, [[-1] * K] * B if cont else [list(range(A))] * B, A, K, cont, max_simulations=S)
    roots.set_tiebreak(0, seed=99)
    noises = rng.dirichlet([0.3] * K, size=B).astype(np.float32)
    out = model.initial_inference(obs, roots)
    roots.prepare_from_inference(0.25, noises, [-1] * B)
    mcts.search(roots, model, out.latent_state, out.reward_hidden_state, [-1] * B)
    node_actions = [roots.get_node_actions(x) for x in range(S + 1)]
    _sampled_replay(model, roots, S, lambda x: node_actions[x], noises, [-1] * B, cont, A_disc=A)
    return "exact replay with the device's own draws: %d roots x %d simulations, K = %d" % (B, S, K)
