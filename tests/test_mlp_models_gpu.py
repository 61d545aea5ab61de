"""Vector-observation (MLP) model family on the device vs the torch fp32 restatement (oracle/torch_models.py) on shared
seeded weights: MuZeroModelMLP (BASELINE configs[0] CartPole shape), EfficientZeroModelMLP, SampledEfficientZeroModelMLP
(BASELINE configs[4] DMC state shape).  Tolerances: pre-h^-1 network outputs 1e-5 (1 + |x|) -- north_star's bound, through parity_record.check --; scalars after h^-1
3e-4 (1 + |x|) (the transform amplifies fp32 rounding of the softmax expectation)."""
import numpy as np
import parity_record
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_models as tm
from oracle import ctree as octree, search as osearch
from lightzero_amd import _lib as L

CFG = dict(num_simulations=25, pb_c_base=19652, pb_c_init=1.25, discount_factor=0.997, value_delta_max=0.01,
           lstm_horizon_len=5, root_noise_weight=0.25, root_dirichlet_alpha=0.3)


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / (1.0 + np.abs(b))))


def _ist(x):
    return tm.InverseScalarTransform()(x).reshape(-1).numpy()


def test_muzero_mlp_search_matches_oracle_pipeline():
    from lightzero_amd.model.muzero_model_mlp import MuZeroModelMLP
    from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree
    from lightzero_amd.mcts.tree_search.mcts_ctree import MuZeroMCTSCtree
    B, A, S = 8, 2, 25
    ref = tm.synthetic_init(tm.MuZeroModelMLP(observation_shape=4, action_space_size=A, latent_state_dim=128), seed=3)
    model = MuZeroModelMLP(observation_shape=4, action_space_size=A, latent_state_dim=128, norm_type='BN').load_state_dict(ref.state_dict())
    obs = torch.randn(B, 4, generator=torch.Generator().manual_seed(1))
    legal = [list(range(A))] * B
    noises = np.random.default_rng(0).dirichlet([0.3] * A, size=B).astype(np.float32).tolist()
    roots = mz_tree.Roots(B, legal, action_space_size=A, max_simulations=S)
    roots.set_tiebreak(0)
    out = model.initial_inference(obs, roots)
    with torch.no_grad():
        ro = ref.initial_inference(obs)
    parity_record.check("mlp_models/%s/site%d" % (__name__.split(".")[-1], 1), {"policy": _rel(out.policy_logits, ro.policy_logits.numpy())})
    assert _rel(out.value, _ist(ro.value)) < 3e-4
    lat = np.zeros((B, 128), np.float32)
    L.check(L.lib().lz_roots_read_latent(roots._h, 0, lat.reshape(-1)))
    parity_record.check("mlp_models/%s/site%d" % (__name__.split(".")[-1], 2), {"latent": _rel(lat, ro.latent_state.numpy())})
    roots.prepare_from_inference(0.25, noises, [-1] * B)
    mcts = MuZeroMCTSCtree(dict(CFG, num_simulations=S))
    L.check(L.lib().lz_roots_enable_trace(roots._h, 1))
    mcts.search(roots, model, out.latent_state, [-1] * B)
    dists, values = roots.get_distributions(), roots.get_values()
    # oracle pipeline: C restatement of the reference tree + torch model
    rec_o = []
    od, ov, _, ol = osearch.mz_forward_collect(octree.mz_tree, ref, obs, legal, noises, [-1] * B, dict(CFG, num_simulations=S),
                                               roots_kwargs=dict(action_space_size=A, max_simulations=S), deterministic=True, record=rec_o)
    same = sum(int(a == b) for a, b in zip(dists, od))
    # BASELINE configs[0]: recorded + every differing root attributed (tests/e2e_common.py)
    import e2e_common
    e2e_common.attribute_and_gate("e2e/mz_mlp_cartpole/B%d_S%d" % (B, S), "mz", octree.mz_tree, dict(CFG, num_simulations=S), A, legal, noises, [-1] * B, ol,
                                  np.asarray(out.policy_logits, np.float32), rec_o, e2e_common.device_records(roots, L.lib(), L, B, A, S),
                                  od, dists, ov, values, gate=(B - 1) / float(B))
    if same == B:
        assert _rel(values, ov) < 3e-4
    # last simulation's network outputs vs torch on the latents the device itself produced
    tr = np.zeros((S, B, 4), np.int32)  # (latent index in search path, last action, search length, virtual to_play)
    L.check(L.lib().lz_roots_read_trace(roots._h, S, tr.reshape(-1)))
    s = S - 1
    ix, la = tr[s, :, 0], tr[s, :, 1]
    pool = np.zeros((S + 1, B, 128), np.float32)
    for k in range(S + 1):
        L.check(L.lib().lz_roots_read_latent(roots._h, k, pool[k].reshape(-1)))
    with torch.no_grad():
        r = ref.recurrent_inference(torch.from_numpy(pool[ix, np.arange(B)]), torch.from_numpy(la).long())
    vp = np.zeros(B, np.float32); v = np.zeros(B, np.float32); lg = np.zeros((B, A), np.float32)
    L.check(L.lib().lz_roots_read_sim_outputs(roots._h, s + 1, vp, v, lg.reshape(-1)))
    parity_record.check("mlp_models/%s/site%d" % (__name__.split(".")[-1], 3), {"latent": _rel(pool[s + 1], r.latent_state.numpy())})
    parity_record.check("mlp_models/%s/site%d" % (__name__.split(".")[-1], 4), {"policy": _rel(lg, r.policy_logits.numpy())})
    assert _rel(v, _ist(r.value)) < 3e-4 and _rel(vp, _ist(r.reward)) < 3e-4


@pytest.mark.parametrize("res", [False, True])
def test_efficientzero_mlp_search_matches_oracle_pipeline(res):
    from lightzero_amd.model.efficientzero_model_mlp import EfficientZeroModelMLP
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    from lightzero_amd.mcts.tree_search.mcts_ctree import EfficientZeroMCTSCtree
    B, A, S, Ld, H = 24, 3, 30, 128, 128
    ref = tm.synthetic_init(tm.EfficientZeroModelMLP(observation_shape=6, action_space_size=A, lstm_hidden_size=H, latent_state_dim=Ld,
                                                     res_connection_in_dynamics=res), seed=5)
    model = EfficientZeroModelMLP(observation_shape=6, action_space_size=A, lstm_hidden_size=H, latent_state_dim=Ld,
                                  res_connection_in_dynamics=res).load_state_dict(ref.state_dict())
    obs = torch.randn(B, 6, generator=torch.Generator().manual_seed(2))
    legal = [list(range(A))] * B
    noises = np.random.default_rng(1).dirichlet([0.3] * A, size=B).astype(np.float32).tolist()
    roots = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S)
    roots.set_tiebreak(0)
    out = model.initial_inference(obs, roots)
    with torch.no_grad():
        ro = ref.initial_inference(obs)
    parity_record.check("mlp_models/%s/site%d" % (__name__.split(".")[-1], 5), {"policy": _rel(out.policy_logits, ro.policy_logits.numpy())})
    roots.prepare_from_inference(0.25, noises, [-1] * B)
    cfg = dict(CFG, num_simulations=S)
    EfficientZeroMCTSCtree(cfg).search(roots, model, out.latent_state, out.reward_hidden_state, [-1] * B)
    od, ov, _, _ = osearch.ez_forward_collect(octree.ez_tree, ref, obs, legal, noises, [-1] * B, cfg,
                                              roots_kwargs=dict(action_space_size=A, max_simulations=S))
    dists = roots.get_distributions()
    same = sum(int(a == b) for a, b in zip(dists, od))
    assert same >= B - 1, "only %d / %d visit-count distributions identical" % (same, B)
    hh = np.zeros((B, H), np.float32); cc = np.zeros((B, H), np.float32)
    L.check(L.lib().lz_roots_read_hidden(roots._h, 1, hh.reshape(-1), cc.reshape(-1)))
    assert np.isfinite(hh).all() and np.abs(hh).max() > 0


def test_efficientzero_mlp_state_norm_and_scalar_heads_search_matches_oracle_pipeline():
    """state_norm=True and categorical_distribution=False (efficientzero_model_mlp.py:32,34; refused until round 6): the fused search on the
    engine model against the whole oracle pipeline -- the torch restatement (bit-equal to the reference module, tests/test_torch_models_vs_reference.py)
    under the restated EfficientZeroMCTSCtree.search with the policy's InverseScalarTransform(support, categorical_distribution=False)."""
    from lightzero_amd.model.efficientzero_model_mlp import EfficientZeroModelMLP
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    from lightzero_amd.mcts.tree_search.mcts_ctree import EfficientZeroMCTSCtree
    B, A, S, Ld, H = 24, 3, 30, 128, 128
    kw = dict(observation_shape=6, action_space_size=A, lstm_hidden_size=H, latent_state_dim=Ld, state_norm=True, categorical_distribution=False)
    ref = tm.synthetic_init(tm.EfficientZeroModelMLP(**kw), seed=7)
    model = EfficientZeroModelMLP(**kw).load_state_dict(ref.state_dict())
    obs = torch.randn(B, 6, generator=torch.Generator().manual_seed(4))
    legal = [list(range(A))] * B
    noises = np.random.default_rng(2).dirichlet([0.3] * A, size=B).astype(np.float32).tolist()
    roots = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S)
    roots.set_tiebreak(0)
    out = model.initial_inference(obs, roots)
    with torch.no_grad():
        ro = ref.initial_inference(obs)
    lat = np.zeros((B, Ld), np.float32)
    L.check(L.lib().lz_roots_read_latent(roots._h, 0, lat.reshape(-1)))
    assert lat.min() == 0.0 and lat.max() == 1.0   # every row renormalised to [0, 1]
    parity_record.check("mlp_models/%s/site%d" % (__name__.split(".")[-1], 7), {"latent": _rel(lat, ro.latent_state.numpy()),
                                                                              "policy": _rel(out.policy_logits, ro.policy_logits.numpy())})
    ist = tm.InverseScalarTransform(categorical_distribution=False)
    assert _rel(out.value, ist(ro.value).reshape(-1).numpy()) < 3e-4
    roots.prepare_from_inference(0.25, noises, [-1] * B)
    cfg = dict(CFG, num_simulations=S, categorical_distribution=False)
    EfficientZeroMCTSCtree(cfg).search(roots, model, out.latent_state, out.reward_hidden_state, [-1] * B)
    od, ov, _, _ = osearch.ez_forward_collect(octree.ez_tree, ref, obs, legal, noises, [-1] * B, cfg,
                                              roots_kwargs=dict(action_space_size=A, max_simulations=S))
    dists, values = roots.get_distributions(), roots.get_values()
    same = sum(int(a == b) for a, b in zip(dists, od))
    assert same >= B - 1, "only %d / %d visit-count distributions identical" % (same, B)
    if same == B:
        assert _rel(values, ov) < 3e-4


def test_sampled_mlp_fused_search_matches_oracle_pipeline():
    """BASELINE configs[4] shape: obs 5, action dim 1, K = 20, latent 256, LSTM 512, LN + GELU, 50 simulations.
    The oracle pipeline's draws are injected into the fused device search: visit counts must be identical."""
    from lightzero_amd.model.sampled_efficientzero_model_mlp import SampledEfficientZeroModelMLP
    from lightzero_amd.mcts.tree_search.mcts_ctree import SampledEfficientZeroMCTSCtree
    B, D, K, S = 32, 1, 20, 50
    ref = tm.synthetic_init(tm.SampledEfficientZeroModelMLP(observation_shape=5, action_space_size=D, num_of_sampled_actions=K), seed=7)
    model = SampledEfficientZeroModelMLP(observation_shape=5, action_space_size=D, continuous_action_space=True,
                                         num_of_sampled_actions=K).load_state_dict(ref.state_dict())
    obs = torch.randn(B, 5, generator=torch.Generator().manual_seed(3))
    cfg = dict(CFG, num_simulations=S, model=dict(action_space_size=D, num_of_sampled_actions=K, continuous_action_space=True))
    with torch.no_grad():
        o = ref.initial_inference(obs)
    noises = np.random.default_rng(0).dirichlet([0.3] * K, size=B).astype(np.float32).tolist()
    oroots = octree.ezs_tree.Roots(B, [[-1] * K] * B, D, K, True, max_simulations=S)
    oroots.set_clock(77)
    oroots.prepare(0.25, noises, [0.] * B, o.policy_logits.numpy().tolist(), [-1] * B)
    osearch.sez_search(octree.ezs_tree, oroots, ref, o.latent_state.numpy(),
                       (o.reward_hidden_state[0].numpy(), o.reward_hidden_state[1].numpy()), [-1] * B, cfg)
    draws = np.stack([np.asarray(oroots.get_sampled_actions(e), np.float32).reshape(B, K, D) for e in range(S + 1)])
    mcts = SampledEfficientZeroMCTSCtree(cfg)
    roots = mcts.roots(B, [[-1] * K] * B, D, K, True, max_simulations=S)
    roots.set_tiebreak(0)
    out = model.initial_inference(obs, roots)
    parity_record.check("mlp_models/%s/site%d" % (__name__.split(".")[-1], 6), {"policy": _rel(out.policy_logits, o.policy_logits.numpy())})
    assert _rel(out.value, _ist(o.value)) < 3e-4
    roots.set_given_records(draws)
    roots.prepare_from_inference(0.25, noises, [-1] * B)
    mcts.search(roots, model, out.latent_state, out.reward_hidden_state, [-1] * B)
    od = oroots.get_distributions()
    dists = roots.get_distributions()
    same = sum(int(a == b) for a, b in zip(dists, od))
    parity_record.record("e2e/sez_mlp_dmc/B%d_S%d" % (B, S), {}, extra=dict(roots=B, identical_visit_distributions=same, identical_fraction=same / float(B), gate=(B - 1) / float(B),
                         note="BASELINE configs[4] shape; the oracle pipeline's draws injected into the fused device search"))
    assert same >= B - 1, "only %d / %d visit-count distributions identical" % (same, B)
    assert np.array_equal(np.asarray(roots.get_sampled_actions(), np.float32), draws[0])
    # device-side sampling: a second search without injected draws still spends every simulation and replays from the graph
    roots.set_given_records(None)
    for _ in range(2):
        out = model.initial_inference(obs, roots)
        roots.prepare_from_inference(0.25, noises, [-1] * B)
        mcts.search(roots, model, out.latent_state, out.reward_hidden_state, [-1] * B)
        d2 = np.asarray(roots.get_distributions())
        assert (d2.sum(1) >= S).all()  # > S only where two sampled actions share a child
        acts = np.asarray(roots.get_sampled_actions())
        assert np.all(np.abs(acts) <= 1.0) and acts.std() > 0


def test_policies_on_mlp_engine_models_full_size():
    """BASELINE configs[0] (CartPole MuZero MLP, 8 envs x 25 sims) and configs[4] (Sampled EfficientZero, DMC state obs,
    K = 20, 256 envs x 50 sims) through the policy surface; size-independent properties at full size."""
    from lightzero_amd.model.muzero_model_mlp import MuZeroModelMLP
    from lightzero_amd.model.sampled_efficientzero_model_mlp import SampledEfficientZeroModelMLP
    from lightzero_amd.policy.muzero import MuZeroPolicy
    from lightzero_amd.policy.sampled_efficientzero import SampledEfficientZeroPolicy
    # configs[0]
    ref = tm.synthetic_init(tm.MuZeroModelMLP(observation_shape=4, action_space_size=2, latent_state_dim=128), seed=0)
    model = MuZeroModelMLP(observation_shape=4, action_space_size=2, latent_state_dim=128).load_state_dict(ref.state_dict())
    pol = MuZeroPolicy(dict(num_simulations=25, discount_factor=0.997, mcts_tiebreak="first"), model)
    obs = torch.randn(8, 4, generator=torch.Generator().manual_seed(0))
    out = pol._forward_collect(obs, action_mask=[np.ones(2)] * 8, temperature=1.0, to_play=[-1] * 8)
    assert all(sum(out[i]["visit_count_distributions"]) == 25 and out[i]["action"] in (0, 1) for i in range(8))
    ev1 = pol._forward_eval(obs, action_mask=[np.ones(2)] * 8, to_play=[-1] * 8)
    ev2 = pol._forward_eval(obs, action_mask=[np.ones(2)] * 8, to_play=[-1] * 8)
    assert all(ev1[i]["visit_count_distributions"] == ev2[i]["visit_count_distributions"] for i in range(8))
    # configs[4]
    B, D, K, S = 256, 1, 20, 50
    ref = tm.synthetic_init(tm.SampledEfficientZeroModelMLP(observation_shape=5, action_space_size=D, num_of_sampled_actions=K), seed=1)
    model = SampledEfficientZeroModelMLP(observation_shape=5, action_space_size=D, continuous_action_space=True,
                                         num_of_sampled_actions=K).load_state_dict(ref.state_dict())
    cfg = dict(num_simulations=S, discount_factor=0.997, lstm_horizon_len=5,
               model=dict(action_space_size=D, num_of_sampled_actions=K, continuous_action_space=True))
    pol = SampledEfficientZeroPolicy(cfg, model)
    obs = torch.randn(B, 5, generator=torch.Generator().manual_seed(4)).cuda()
    import time
    out = pol._forward_collect(obs, temperature=1.0, to_play=[-1] * B)
    t0 = time.perf_counter()
    for _ in range(5):
        out = pol._forward_collect(obs, temperature=1.0, to_play=[-1] * B)
    dt = (time.perf_counter() - t0) / 5
    print("configs[4] policy forward: %.2f ms / env-step batch of %d (%.0f env-steps/s incl. host glue)" % (dt * 1e3, B, B / dt))
    assert len(out) == B
    for i in range(B):
        o = out[i]
        # sampled actions with the same "%f" key share one child (cnode.cpp:55-110), so the counts of distinct keys add up to S
        keys = np.rint(o["root_sampled_actions"].astype(np.float64) * 1e6).astype(np.int64)
        first = [j for j in range(K) if not any((keys[j] == keys[q]).all() for q in range(j))]
        assert sum(o["visit_count_distributions"][j] for j in first) == S and len(o["visit_count_distributions"]) == K
        assert o["root_sampled_actions"].shape == (K, D) and np.all(np.abs(o["root_sampled_actions"]) <= 1.0)
        assert any(np.array_equal(o["action"], a) for a in o["root_sampled_actions"])
        assert np.isfinite(o["searched_value"]) and np.isfinite(o["predicted_value"])
    # sampled actions follow the root policy: tanh(N(mu, sigma)) -> atanh(a) has mean ~ mu
    mu = np.array([out[i]["predicted_policy_logits"][0] for i in range(B)])
    a = np.stack([out[i]["root_sampled_actions"][:, 0] for i in range(B)])
    z = np.arctanh(np.clip(a, -0.999999, 0.999999))
    sig = np.array([out[i]["predicted_policy_logits"][1] for i in range(B)])
    assert np.abs(((z - mu[:, None]) / sig[:, None]).mean()) < 0.1
    ev = pol._forward_eval(obs, to_play=[-1] * B)
    assert all(sum(ev[i]["visit_count_distributions"]) >= S for i in range(B))


def test_discrete_sampled_efficientzero_fused_and_policy():
    """Sampled EfficientZero on a DISCRETE action space (cartpole_sampled_efficientzero_config.py shape: A = 2, K = 2 is degenerate;
    use A = 7, K = 4): engine model + fused search with the oracle's draws injected reproduces the oracle pipeline; policy surface."""
    from lightzero_amd.model.sampled_efficientzero_model_mlp import SampledEfficientZeroModelMLP
    from lightzero_amd.mcts.tree_search.mcts_ctree import SampledEfficientZeroMCTSCtree
    from lightzero_amd.policy.sampled_efficientzero import SampledEfficientZeroPolicy
    B, A, K, S = 32, 7, 4, 30
    ref = tm.synthetic_init(tm.SampledEfficientZeroModelMLP(observation_shape=6, action_space_size=A, continuous_action_space=False,
                                                            num_of_sampled_actions=K), seed=11)
    model = SampledEfficientZeroModelMLP(observation_shape=6, action_space_size=A, continuous_action_space=False,
                                         num_of_sampled_actions=K).load_state_dict(ref.state_dict())
    obs = torch.randn(B, 6, generator=torch.Generator().manual_seed(8))
    cfg = dict(CFG, num_simulations=S, model=dict(action_space_size=A, num_of_sampled_actions=K, continuous_action_space=False))
    with torch.no_grad():
        o = ref.initial_inference(obs)
    noises = np.random.default_rng(0).dirichlet([0.3] * K, size=B).astype(np.float32).tolist()
    oroots = octree.ezs_tree.Roots(B, [list(range(A))] * B, A, K, False, max_simulations=S)
    oroots.set_clock(99)
    oroots.prepare(0.25, noises, [0.] * B, o.policy_logits.numpy().tolist(), [-1] * B)

    class _Disc(object):  # the oracle driver feeds float last_actions; a discrete model wants indices
        def eval(self):
            return self

        def recurrent_inference(self, z, hc, a):
            return ref.recurrent_inference(z, hc, a.reshape(-1).long())
    osearch.sez_search(octree.ezs_tree, oroots, _Disc(), o.latent_state.numpy(),
                       (o.reward_hidden_state[0].numpy(), o.reward_hidden_state[1].numpy()), [-1] * B, cfg)
    draws = np.stack([np.asarray(oroots.get_sampled_actions(e), np.float32).reshape(B, K, 1) for e in range(S + 1)])
    mcts = SampledEfficientZeroMCTSCtree(cfg)
    roots = mcts.roots(B, [list(range(A))] * B, A, K, False, max_simulations=S)
    roots.set_tiebreak(0)
    out = model.initial_inference(obs, roots)
    parity_record.check("mlp_models/%s/site%d" % (__name__.split(".")[-1], 7), {"policy": _rel(out.policy_logits, o.policy_logits.numpy())})
    roots.set_given_records(draws)
    roots.prepare_from_inference(0.25, noises, [-1] * B)
    mcts.search(roots, model, out.latent_state, out.reward_hidden_state, [-1] * B)
    same = sum(int(a == b) for a, b in zip(roots.get_distributions(), oroots.get_distributions()))
    assert same >= B - 2, "only %d / %d visit-count distributions identical" % (same, B)
    roots.set_given_records(None)
    policy = SampledEfficientZeroPolicy(cfg, model)
    res = policy._forward_collect(obs, temperature=1.0, to_play=[-1] * B)
    for i in range(B):
        assert isinstance(res[i]["action"], int) and 0 <= res[i]["action"] < A
        acts = res[i]["root_sampled_actions"].reshape(-1)
        assert len(set(acts.tolist())) == K and sum(res[i]["visit_count_distributions"]) == S
