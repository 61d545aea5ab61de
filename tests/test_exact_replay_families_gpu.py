"""The exact end-to-end gate (tests/test_exact_replay_gpu.py) for every OTHER fused loop that ships: the vector-observation
models (BASELINE configs[0] MuZeroModelMLP 8 x 25; EfficientZeroModelMLP), Sampled EfficientZero at BASELINE configs[4] full size
(256 roots x 50 simulations, K = 20) -- once with the DEVICE's own draws (the production path: the actions every node sampled
on the device are read back and injected into the oracle) and once with the oracle's draws injected into the device
(lz_sroots_set_given) --, Gumbel MuZero (lz_gsearch, 64 x 50, A = 18) and ReZero (lz_search_with_reuse, EfficientZero and MuZero).

Same bar everywhere: the production launch sequence runs on the device; the device's OWN per-simulation network outputs are
replayed through the CPU tree oracle (oracle/ctree_*.c, each pinned bit-exact to the reference's compiled ctree) and -- where
oracle/_ref is on the box and the module takes recorded inputs -- through the reference's compiled module itself: 100 % of the
roots with identical visit counts, bit-equal root values and min-max statistics, identical per-simulation records."""
import numpy as np
import pytest
import torch

import gumbel_driver as gd
import sampled_driver as sd
import tree_driver as td
from test_exact_replay_gpu import _search_and_replay

pytestmark = pytest.mark.gpu

PB = dict(pb_c_base=19652, pb_c_init=1.25, delta=0.01, horizon=5)


def _sims(roots, S, B, PW):
    from lightzero_amd import _lib as L
    sims = []
    for s in range(1, S + 1):
        vp = np.zeros(B, np.float32); v = np.zeros(B, np.float32); lg = np.zeros((B, PW), np.float32)
        L.check(L.lib().lz_roots_read_sim_outputs(roots._h, s, vp, v, lg.reshape(-1)))
        sims.append(dict(vp=vp, v=v, logits=lg))
    return sims


# ---------------------------------------------------------------------------------------------------------------- MLP models
def test_configs0_cartpole_muzero_mlp_replays_exactly():
    """BASELINE configs[0]: CartPole MuZeroModelMLP, 8 roots x 25 simulations -- full size -- and the same model at 256 roots"""
    from oracle import torch_models as tm
    from lightzero_amd.model.muzero_model_mlp import MuZeroModelMLP
    from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree
    A = 2
    sd_ = tm.synthetic_init(tm.MuZeroModelMLP(observation_shape=4, action_space_size=A, latent_state_dim=128), seed=3).state_dict()
    model = MuZeroModelMLP(observation_shape=4, action_space_size=A, latent_state_dim=128).load_state_dict(sd_)
    for B, S, seed in ((8, 25, 1), (256, 50, 2)):
        obs = torch.randn(B, 4, generator=torch.Generator().manual_seed(seed)).cuda().contiguous()
        rng = np.random.default_rng(seed)
        noises = [rng.dirichlet([0.3] * A).astype(np.float32).tolist() for _ in range(B)]
        legal = [list(range(A))] * B
        roots = mz_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
        roots.set_tiebreak(0)
        _search_and_replay("mz", model, roots, obs, legal, [-1] * B, noises, S, 0.997, trace=True)


@pytest.mark.parametrize("res", [False, True])
def test_efficientzero_mlp_replays_exactly(res):
    from oracle import torch_models as tm
    from lightzero_amd.model.efficientzero_model_mlp import EfficientZeroModelMLP
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    B, A, S = 96, 3, 50
    kw = dict(observation_shape=6, action_space_size=A, lstm_hidden_size=128, latent_state_dim=128, res_connection_in_dynamics=res)
    model = EfficientZeroModelMLP(**kw).load_state_dict(tm.synthetic_init(tm.EfficientZeroModelMLP(**kw), seed=5).state_dict())
    obs = torch.randn(B, 6, generator=torch.Generator().manual_seed(2)).cuda().contiguous()
    rng = np.random.default_rng(7)
    legal = []
    for _ in range(B):
        m = rng.random(A) < 0.8
        m[rng.integers(0, A)] = True
        legal.append(np.nonzero(m)[0].tolist())
    noises = [rng.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal]
    roots = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
    roots.set_tiebreak(0)
    _search_and_replay("ez", model, roots, obs, legal, [-1] * B, noises, S, 0.997, trace=True)


def test_lunarlander_sized_efficientzero_mlp_replays_exactly():
    """the sizes of the reference's LunarLander / BipedalWalker / MiniGrid EfficientZero configs (latent_state_dim 256, lstm_hidden_size
    256, zoo/box2d/lunarlander/config/lunarlander_disc_efficientzero_config.py): 256 roots x 50 simulations"""
    from oracle import torch_models as tm
    from lightzero_amd.model.efficientzero_model_mlp import EfficientZeroModelMLP
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    B, A, S = 256, 4, 50
    kw = dict(observation_shape=8, action_space_size=A, lstm_hidden_size=256, latent_state_dim=256)
    model = EfficientZeroModelMLP(**kw).load_state_dict(tm.synthetic_init(tm.EfficientZeroModelMLP(**kw), seed=9).state_dict())
    obs = torch.randn(B, 8, generator=torch.Generator().manual_seed(4)).cuda().contiguous()
    rng = np.random.default_rng(8)
    legal = [list(range(A))] * B
    noises = [rng.dirichlet([0.3] * A).astype(np.float32).tolist() for _ in range(B)]
    roots = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
    roots.set_tiebreak(0)
    _search_and_replay("ez", model, roots, obs, legal, [-1] * B, noises, S, 0.997, trace=True)


# ------------------------------------------------------------------------------------------------------ Sampled EfficientZero
def _sampled_replay(model, roots, S, draws_of, noises, to_play, continuous, A_disc=None):
    """device results + per-simulation outputs -> the C oracle with `draws_of(record)` injected at every expand"""
    from oracle import ctree as octree
    from lightzero_amd import _lib as L
    lib = L.lib()
    B, K, D = roots.root_num, roots.K, roots.D
    PW = model._policy_width
    pred = np.zeros(B, np.float32); pol0 = np.zeros((B, PW), np.float32)
    L.check(lib.lz_roots_get_root_outputs(roots._h, pred, pol0.reshape(-1)))
    sims = _sims(roots, S, B, PW)
    c = dict(B=B, D=D, K=K, S=S, discount=0.997, noise_w=0.25, noises=noises, root_vp=np.zeros(B, np.float32), root_policy=pol0,
             to_play_list=list(to_play), sims=[dict(vp=x["vp"], v=x["v"], policy=x["logits"]) for x in sims], **PB)
    if not continuous:
        c["A"] = A_disc

    def mk():
        r = octree.ezs_tree.Roots(B, [[-1] * K] * B, A_disc if not continuous else D, K, continuous, max_simulations=S)
        r.set_tiebreak(0)
        return r
    ora = sd.run_tree(octree.ezs_tree, c, mk, before_expand=lambda r, e: setattr(r, "given", draws_of(e)))
    d_dist = np.asarray(roots.get_distributions(), np.int32)
    same = int((ora["distributions"] == d_dist).all(1).sum())
    assert same == B, "only %d / %d roots have identical visit counts" % (same, B)
    assert np.array_equal(ora["values"].view(np.uint32), np.asarray(roots.get_values(), np.float32).view(np.uint32)), "root values not bit-equal"
    return ora, sims


def _sez_model(continuous, A, K, obs_dim, seed):
    from oracle import torch_models as tm
    from lightzero_amd.model.sampled_efficientzero_model_mlp import SampledEfficientZeroModelMLP
    ref = tm.synthetic_init(tm.SampledEfficientZeroModelMLP(observation_shape=obs_dim, action_space_size=A, continuous_action_space=continuous,
                                                            num_of_sampled_actions=K), seed=seed)
    return SampledEfficientZeroModelMLP(observation_shape=obs_dim, action_space_size=A, continuous_action_space=continuous,
                                        num_of_sampled_actions=K).load_state_dict(ref.state_dict())


@pytest.mark.parametrize("continuous", [True, False])
def test_configs4_sampled_efficientzero_full_size_device_draws_replay_exactly(continuous):
    """BASELINE configs[4]: DMC state obs 5, action dim 1, K = 20, 256 roots x 50 simulations, LN + GELU MLPs + LSTM 512 --
    the PRODUCTION path: every node's K actions are drawn on the device inside the captured search graph.  The actions each
    expanded node holds are read back (lz_sroots_get_node_actions) and injected into the oracle tree; with them and the
    device's own network outputs the oracle must reproduce the search exactly (100 % of the roots, bit-equal values, identical
    records).  Discrete variant: A = 9 actions, K = 5 sampled without replacement."""
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.tree_search.mcts_ctree import SampledEfficientZeroMCTSCtree
    B, S = 256, 50
    D, K, A = (1, 20, 1) if continuous else (1, 5, 9)
    model = _sez_model(continuous, A, K, 5, seed=21)
    cfg = dict(num_simulations=S, pb_c_base=19652, pb_c_init=1.25, discount_factor=0.997, value_delta_max=0.01, lstm_horizon_len=5,
               root_noise_weight=0.25, model=dict(action_space_size=A, num_of_sampled_actions=K, continuous_action_space=continuous))
    mcts = SampledEfficientZeroMCTSCtree(cfg)
    legal = [[-1] * K] * B if continuous else [list(range(A))] * B
    roots = mcts.roots(B, legal, A, K, continuous, max_simulations=S)
    roots.set_tiebreak(0, seed=1234)
    obs = torch.randn(B, 5, generator=torch.Generator().manual_seed(31)).cuda().contiguous()
    noises = np.random.default_rng(5).dirichlet([0.3] * K, size=B).astype(np.float32)
    out = model.initial_inference(obs, roots)
    L.check(L.lib().lz_roots_enable_trace(roots._h, 1))
    roots.prepare_from_inference(0.25, noises, [-1] * B)
    mcts.search(roots, model, out.latent_state, out.reward_hidden_state, [-1] * B)
    node_actions = [roots.get_node_actions(e) for e in range(S + 1)]
    assert all(np.isfinite(a).all() for a in node_actions)
    if continuous:
        assert np.abs(node_actions[0]).max() <= 1.0 and np.std(node_actions[0]) > 0
    ora, _ = _sampled_replay(model, roots, S, lambda e: node_actions[e], noises, [-1] * B, continuous, A_disc=A)
    tr = np.zeros((S, B, 4), np.int32)
    L.check(L.lib().lz_roots_read_trace(roots._h, S, tr.reshape(-1)))
    assert np.array_equal(ora["records"][:, :, [0, 2]], tr[:, :, [0, 2]]), "per-simulation (parent slot, search length) records differ"
    assert np.array_equal(ora["root_actions"].view(np.uint32), node_actions[0].view(np.uint32))
    # the same roots again: a second search replays the captured graph with fresh draws (epoch bumped by prepare)
    out = model.initial_inference(obs, roots)
    roots.prepare_from_inference(0.25, noises, [-1] * B)
    mcts.search(roots, model, out.latent_state, out.reward_hidden_state, [-1] * B)
    node_actions2 = [roots.get_node_actions(e) for e in range(S + 1)]
    assert not np.array_equal(node_actions2[0], node_actions[0]), "the second search drew the same root actions"
    _sampled_replay(model, roots, S, lambda e: node_actions2[e], noises, [-1] * B, continuous, A_disc=A)


@pytest.mark.parametrize("obs_dim,A,K", [(11, 125, 20), (24, 256, 20)])
def test_discretised_sampled_efficientzero_configs_beyond_64_actions_replay_exactly(obs_dim, A, K):
    """The shipped discretised-action Sampled EfficientZero presets whose action space exceeds one 64-lane chunk:
    zoo/mujoco/config/mujoco_disc_sampled_efficientzero_config.py (Hopper: 5^3 = 125 actions, K = 20) and
    zoo/box2d/bipedalwalker/config/bipedalwalker_cont_disc_sampled_efficientzero_config.py (4^4 = 256 actions, K = 20).  Every node's K
    actions are drawn WITHOUT replacement on the device inside the captured search graph (four actions per lane); read back and injected
    into the oracle sampled tree they must reproduce the search exactly -- records, visit counts, bit-equal values."""
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.tree_search.mcts_ctree import SampledEfficientZeroMCTSCtree
    B, S = 128, 50
    model = _sez_model(False, A, K, obs_dim, seed=23)
    cfg = dict(num_simulations=S, pb_c_base=19652, pb_c_init=1.25, discount_factor=0.997, value_delta_max=0.01, lstm_horizon_len=5,
               root_noise_weight=0.25, model=dict(action_space_size=A, num_of_sampled_actions=K, continuous_action_space=False))
    mcts = SampledEfficientZeroMCTSCtree(cfg)
    roots = mcts.roots(B, [list(range(A))] * B, A, K, False, max_simulations=S)
    roots.set_tiebreak(0, seed=4321)
    obs = torch.randn(B, obs_dim, generator=torch.Generator().manual_seed(33)).cuda().contiguous()
    noises = np.random.default_rng(6).dirichlet([0.3] * K, size=B).astype(np.float32)
    out = model.initial_inference(obs, roots)
    L.check(L.lib().lz_roots_enable_trace(roots._h, 1))
    roots.prepare_from_inference(0.25, noises, [-1] * B)
    mcts.search(roots, model, out.latent_state, out.reward_hidden_state, [-1] * B)
    node_actions = [roots.get_node_actions(e) for e in range(S + 1)]
    for a in node_actions:   # K DISTINCT action indices per node, inside the action space, beyond the first chunk somewhere
        a = np.asarray(a).reshape(B, K)
        assert np.array_equal(a, np.rint(a)) and a.min() >= 0 and a.max() < A
        assert all(len(set(row.tolist())) == K for row in a)
    assert max(np.asarray(a).max() for a in node_actions) >= 64
    ora, _ = _sampled_replay(model, roots, S, lambda e: node_actions[e], noises, [-1] * B, False, A_disc=A)
    tr = np.zeros((S, B, 4), np.int32)
    L.check(L.lib().lz_roots_read_trace(roots._h, S, tr.reshape(-1)))
    assert np.array_equal(ora["records"][:, :, [0, 2]], tr[:, :, [0, 2]]), "per-simulation (parent slot, search length) records differ"


def test_conv_sampled_efficientzero_atari_config_replays_exactly():
    """The convolutional Sampled EfficientZero as the reference ships it for Atari (zoo/atari/config/atari_sampled_efficientzero_config.py:
    4 x 64 x 64 observations, discrete actions, K = 5 sampled without replacement, norm_type='BN', the class defaults GELU / 256-wide heads):
    the fused search on the device (draws inside the captured graph) replays exactly through the oracle sampled tree with the device's own
    draws and network outputs -- visit counts, root values, per-simulation records."""
    from oracle import torch_models as tm
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.tree_search.mcts_ctree import SampledEfficientZeroMCTSCtree
    from lightzero_amd.model.sampled_efficientzero_model import SampledEfficientZeroModel
    B, S, A, K = 96, 40, 6, 5
    kw = dict(observation_shape=(4, 64, 64), action_space_size=A, num_of_sampled_actions=K, downsample=True, continuous_action_space=False, norm_type='BN')
    ref = tm.synthetic_init(tm.SampledEfficientZeroModel(**kw), seed=41)
    model = SampledEfficientZeroModel(**kw).load_state_dict(ref.state_dict())
    cfg = dict(num_simulations=S, pb_c_base=19652, pb_c_init=1.25, discount_factor=0.997, value_delta_max=0.01, lstm_horizon_len=5,
               root_noise_weight=0.25, model=dict(action_space_size=A, num_of_sampled_actions=K, continuous_action_space=False))
    mcts = SampledEfficientZeroMCTSCtree(cfg)
    roots = mcts.roots(B, [list(range(A))] * B, A, K, False, max_simulations=S)
    roots.set_tiebreak(0, seed=99)
    obs = torch.rand(B, 4, 64, 64, generator=torch.Generator().manual_seed(33)).cuda().contiguous()
    noises = np.random.default_rng(6).dirichlet([0.3] * K, size=B).astype(np.float32)
    out = model.initial_inference(obs, roots)
    L.check(L.lib().lz_roots_enable_trace(roots._h, 1))
    roots.prepare_from_inference(0.25, noises, [-1] * B)
    mcts.search(roots, model, out.latent_state, out.reward_hidden_state, [-1] * B)
    node_actions = [roots.get_node_actions(e) for e in range(S + 1)]
    assert all(np.isfinite(a).all() for a in node_actions)
    ora, _ = _sampled_replay(model, roots, S, lambda e: node_actions[e], noises, [-1] * B, False, A_disc=A)
    tr = np.zeros((S, B, 4), np.int32)
    L.check(L.lib().lz_roots_read_trace(roots._h, S, tr.reshape(-1)))
    assert np.array_equal(ora["records"][:, :, [0, 2]], tr[:, :, [0, 2]]), "per-simulation (parent slot, search length) records differ"
    assert np.array_equal(ora["root_actions"].view(np.uint32), node_actions[0].view(np.uint32))
    assert (np.asarray(roots.get_distributions()).sum(1) == S).all()


def test_configs4_sampled_efficientzero_full_size_injected_draws_replay_exactly():
    """the other direction at the same size: draws made by the ORACLE's generator (the reference's minstd_rand0 /
    normal_distribution restated, seeded by set_clock) are injected into the fused device search (lz_sroots_set_given)"""
    from oracle import ctree as octree
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.tree_search.mcts_ctree import SampledEfficientZeroMCTSCtree
    B, D, K, S = 256, 1, 20, 50
    model = _sez_model(True, D, K, 5, seed=22)
    cfg = dict(num_simulations=S, pb_c_base=19652, pb_c_init=1.25, discount_factor=0.997, value_delta_max=0.01, lstm_horizon_len=5,
               root_noise_weight=0.25, model=dict(action_space_size=D, num_of_sampled_actions=K, continuous_action_space=True))
    mcts = SampledEfficientZeroMCTSCtree(cfg)
    roots = mcts.roots(B, [[-1] * K] * B, D, K, True, max_simulations=S)
    roots.set_tiebreak(0)
    obs = torch.randn(B, 5, generator=torch.Generator().manual_seed(32)).cuda().contiguous()
    noises = np.random.default_rng(6).dirichlet([0.3] * K, size=B).astype(np.float32)
    # draws: the oracle's own generator on N(mu, sigma) of an arbitrary (mu | sigma) per record -- any set of post-tanh actions
    # is a valid injection; use the oracle tree's sampler so that the "%f"-key duplicates it produces are the real ones
    rng = np.random.default_rng(8)
    draws = np.tanh(0.3 * rng.standard_normal((S + 1, B, K, D)) + 0.5 * rng.standard_normal((S + 1, B, 1, D))).astype(np.float32)
    draws[3, :, 5] = draws[3, :, 2]    # duplicates: two sampled actions with the same "%f" key share one child (cnode.cpp:55-110)
    draws[0, ::7, 11] = draws[0, ::7, 0]
    out = model.initial_inference(obs, roots)
    roots.set_given_records(draws)
    roots.prepare_from_inference(0.25, noises, [-1] * B)
    mcts.search(roots, model, out.latent_state, out.reward_hidden_state, [-1] * B)
    assert np.array_equal(roots.get_node_actions(0), draws[0]) and np.array_equal(roots.get_node_actions(S), draws[S])
    _sampled_replay(model, roots, S, lambda e: draws[e], noises, [-1] * B, True)
    roots.set_given_records(None)


# --------------------------------------------------------------------------------------------------------------- Gumbel MuZero
def _gumbel_search_and_replay(model, mcts, B, A, S, m, legal, noises, obs, discount=0.997):
    """lz_gsearch on the device, then the device's own network outputs through oracle/ctree_gumbel_oracle.c and the reference's compiled
    gmz_tree: identical records, visit counts, bit-equal root values, improved policies and completed Q-values.  noises None: no root noise."""
    from oracle import build_ref, ctree as octree
    from lightzero_amd import _lib as L
    roots = mcts.roots(B, legal, action_space_size=A, max_simulations=S)
    out = model.initial_inference(obs, roots)
    L.check(L.lib().lz_roots_enable_trace(roots._h, 1))
    if noises is not None:
        roots.prepare_from_inference(0.25, noises, [-1] * B)
    else:
        roots.prepare_from_inference_no_noise([-1] * B)
    mcts.search(roots, model, out.latent_state, [-1] * B)
    pred = np.zeros(B, np.float32); pol0 = np.zeros((B, A), np.float32)
    L.check(L.lib().lz_roots_get_root_outputs(roots._h, pred, pol0.reshape(-1)))
    sims = _sims(roots, S, B, A)
    tr = np.zeros((S, B, 4), np.int32)
    L.check(L.lib().lz_roots_read_trace(roots._h, S, tr.reshape(-1)))
    c = dict(B=B, A=A, S=S, m=m, discount=discount, delta=0.01, noise_w=0.25, legal_list=legal, root_logits=pol0,
             root_reward=np.zeros(B, np.float32), root_value=pred, noises=noises,
             sims=[dict(r=x["vp"], v=x["v"], logits=x["logits"]) for x in sims])
    dev = dict(distributions=np.full((B, A), -1, np.int32), values=np.asarray(roots.get_values(), np.float32),
               policies=np.asarray(roots.get_policies(discount, A), np.float32),
               children_values=np.asarray(roots.get_children_values(discount, A), np.float32))
    for i, d in enumerate(roots.get_distributions()):
        dev["distributions"][i, :len(d)] = d
    mods = [("oracle/ctree_gumbel_oracle.c", octree.gmz_tree, dict(action_space_size=A, max_simulations=S))]
    ref = build_ref.load_gumbel()
    if ref is not None:
        mods.append(("oracle/_ref (the reference's own gmz_tree)", ref, None))
    for name, mod, kw in mods:
        o = gd.run_tree(mod, c, roots_kwargs=kw)
        same = int((o["distributions"] == dev["distributions"]).all(1).sum())
        assert same == B, "%s: only %d / %d roots have identical visit counts" % (name, same, B)
        for k in ("values", "policies", "children_values"):
            assert np.array_equal(o[k].view(np.uint32), dev[k].view(np.uint32)), "%s: %s not bit-equal" % (name, k)
        assert np.array_equal(o["records"][:, :, [0, 2, 3]], tr[:, :, [0, 1, 2]]), "%s: per-simulation records differ" % name
    assert (np.where(dev["distributions"] < 0, 0, dev["distributions"]).sum(1) == S).all()


def test_gumbel_fused_search_replays_exactly():
    """lz_gsearch (GumbelMuZeroMCTSCtree.search, mcts_ctree.py:1067-1172) with the engine MuZero model, 64 roots x 50 simulations,
    A = 18 (the Atari full action set), m = 16 considered actions, ragged legal masks: the device's own network outputs replayed
    through oracle/ctree_gumbel_oracle.c and the reference's compiled gmz_tree -- identical records, visit counts, bit-equal root
    values, improved policies and completed Q-values."""
    from oracle import torch_models as tm
    from lightzero_amd.mcts.tree_search.mcts_ctree import GumbelMuZeroMCTSCtree
    from lightzero_amd.model.muzero_model import MuZeroModel
    B, A, S, m = 64, 18, 50, 16
    model = MuZeroModel(action_space_size=A).load_state_dict(tm.synthetic_init(tm.MuZeroModel(action_space_size=A), seed=41).state_dict())
    cfg = dict(num_simulations=S, discount_factor=0.997, max_num_considered_actions=m, value_delta_max=0.01, root_noise_weight=0.25)
    mcts = GumbelMuZeroMCTSCtree(cfg)
    rng = np.random.default_rng(42)
    legal = []
    for _ in range(B):
        k = rng.random(A) < 0.75
        k[rng.integers(0, A)] = True
        legal.append(np.nonzero(k)[0].tolist())
    noises = [rng.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal]
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(43)).cuda().contiguous()
    for use_noise in (True, False):
        _gumbel_search_and_replay(model, mcts, B, A, S, m, legal, noises if use_noise else None, obs)


# ----------------------------------------------------------------------------------------------------------------------- ReZero
def _reuse_replay(variant, model, roots, mcts, obs, legal, to_play, noises, S, discount, true_action, reuse_value):
    from oracle import build_ref, ctree as octree
    from lightzero_amd import _lib as L
    lib = L.lib()
    B, A = roots.num, model.action_space_size
    out = model.initial_inference(obs, roots)
    L.check(lib.lz_roots_enable_trace(roots._h, 1))
    roots.prepare_from_inference(0.25, noises, to_play)
    if variant == "ez":
        length, avg = mcts.search_with_reuse(roots, model, out.latent_state, out.reward_hidden_state, to_play, true_action, reuse_value)
    else:
        length, avg = mcts.search_with_reuse(roots, model, out.latent_state, to_play, true_action, reuse_value)
    d_dist = roots.get_distributions()
    d_val = np.asarray(roots.get_values(), np.float32)
    d_mm = roots.get_minmax()
    sims = _sims(roots, S, B, A)
    tr = np.zeros((S, B, 4), np.int32)
    L.check(lib.lz_roots_read_trace(roots._h, S, tr.reshape(-1)))
    c = dict(variant=variant, B=B, A=A, S=S, legal_list=[list(l) for l in legal], to_play_list=list(to_play), root_logits=out.policy_logits,
             root_vp=np.zeros(B, np.float32), noises=noises, noise_w=0.25, sims=sims, discount=discount,
             true_action=list(true_action), reuse_value=np.asarray(reuse_value, np.float32), **PB)
    mods = [("oracle/ctree_oracle.c", octree.ez_tree if variant == "ez" else octree.mz_tree, dict(action_space_size=A, max_simulations=S))]
    ref = build_ref.load("det")
    if ref:
        mods.append(("oracle/_ref/det (the reference's own ctree)", ref[0] if variant == "ez" else ref[1], None))
    for name, mod, kw in mods:
        o = td.run_tree_reuse(mod, c, roots_kwargs=kw)
        same = sum(int(a == b) for a, b in zip(o["distributions"], d_dist))
        assert same == B, "%s: only %d / %d roots have identical visit-count distributions" % (name, same, B)
        assert np.array_equal(o["values"].view(np.uint32), d_val.view(np.uint32)), "%s: root values not bit-equal" % name
        assert abs(o["inferences"] / S - avg) < 1e-9, "%s: average inference batch %r vs %r" % (name, o["inferences"] / S, avg)
        # records of the roots that went through the network in a simulation: (parent slot, action, search length)
        need = o["records"][:, :, 0] >= 0
        assert np.array_equal(o["records"][:, :, [0, 2, 3]][need], tr[:, :, [0, 1, 2]][need]), "%s: per-simulation records differ" % name
    assert np.isfinite(d_mm).all()
    return avg


def test_rezero_efficientzero_fused_search_with_reuse_replays_exactly():
    """lz_search_with_reuse (EfficientZeroMCTSCtree.search_with_reuse, mcts_ctree.py:878-1002) with the engine conv model, 128 x 50"""
    from oracle import torch_models as tm
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    from lightzero_amd.mcts.tree_search.mcts_ctree import EfficientZeroMCTSCtree
    B, A, S = 128, 6, 50
    model = EfficientZeroModel(action_space_size=A).load_state_dict(tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A), seed=51).state_dict())
    cfg = dict(num_simulations=S, pb_c_base=19652, pb_c_init=1.25, discount_factor=0.997, value_delta_max=0.01, lstm_horizon_len=5)
    mcts = EfficientZeroMCTSCtree(cfg)
    rng = np.random.default_rng(52)
    legal = [list(range(A))] * B
    noises = [rng.dirichlet([0.3] * A).astype(np.float32).tolist() for _ in range(B)]
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(53)).cuda().contiguous()
    roots = mcts.roots(B, legal, action_space_size=A, max_simulations=S)
    roots.set_tiebreak(0)
    true_action = rng.integers(0, A, size=B).tolist()
    reuse_value = rng.standard_normal(B).astype(np.float32).tolist()
    avg = _reuse_replay("ez", model, roots, mcts, obs, legal, [-1] * B, noises, S, 0.997, true_action, reuse_value)
    assert 0 < avg < B, "no root ever reused its true action (average inference batch %r)" % avg


def test_rezero_muzero_fused_search_with_reuse_replays_exactly():
    """MuZeroMCTSCtree.search_with_reuse (mcts_ctree.py:370-470): Go 9x9, two players, ragged legal masks, 48 x 60"""
    from oracle import torch_models as tm
    from lightzero_amd.model.muzero_model import MuZeroModel
    from lightzero_amd.mcts.tree_search.mcts_ctree import MuZeroMCTSCtree
    B, A, S = 48, 82, 60
    kw = dict(observation_shape=(17, 9, 9), downsample=False)
    model = MuZeroModel(action_space_size=A, **kw).load_state_dict(tm.synthetic_init(tm.MuZeroModel(action_space_size=A, **kw), seed=54).state_dict())
    cfg = dict(num_simulations=S, pb_c_base=19652, pb_c_init=1.25, discount_factor=1.0, value_delta_max=0.01, env_type="board_games")
    mcts = MuZeroMCTSCtree(cfg)
    rng = np.random.default_rng(55)
    obs = (torch.rand(B, 17, 9, 9, generator=torch.Generator().manual_seed(56)) < 0.3).float().cuda().contiguous()
    legal = []
    for _ in range(B):
        k = rng.random(A) < 0.7
        k[A - 1] = True
        legal.append(np.nonzero(k)[0].tolist())
    to_play = rng.integers(1, 3, size=B).tolist()
    noises = [rng.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal]
    roots = mcts.roots(B, legal, action_space_size=A, max_simulations=S)
    roots.set_tiebreak(0)
    true_action = [int(l[rng.integers(0, len(l))]) for l in legal]
    reuse_value = rng.standard_normal(B).astype(np.float32).tolist()
    _reuse_replay("mz", model, roots, mcts, obs, legal, to_play, noises, S, 1.0, true_action, reuse_value)
