"""GPU: randomised sweep of the FUSED search loops through the exact replay gate (tests/test_exact_replay_gpu.py: the device's own
network outputs of every simulation replayed through the oracle trees -> identical visit counts, bit-equal root values and min-max
statistics, identical per-simulation selection records): EfficientZero Atari models (split heads, tree step in the chain launch) and
two-player MuZero board models at random batch sizes, action counts, simulation counts, ragged legal lists, noise on / off."""
import numpy as np
import pytest
import torch

from test_exact_replay_gpu import _ez_model, _mz_model, _search_and_replay

# LZ_FUZZ_SEED_OFFSET=n shifts every seeded sweep of this file to seeds n .. n + count - 1 (ad-hoc wider sweeps; the committed suite runs 0)
_OFF = int(__import__("os").environ.get("LZ_FUZZ_SEED_OFFSET", "0"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(_OFF, _OFF + 10))
def test_random_efficientzero_search_replays_exactly(seed):
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    r = np.random.default_rng(1200 + seed)
    B, A, S = int(r.integers(1, 300)), int(r.integers(2, 19)), int(r.integers(1, 70))
    hw = int(r.choice([96, 64]))
    kw = dict(observation_shape=(4, hw, hw))
    if hw == 64:
        kw.update(reward_support_range=(-50., 51., 1.), value_support_range=(-50., 51., 1.))
    model = _ez_model(A, seed=seed, **kw)
    obs = torch.rand(B, 4, hw, hw, generator=torch.Generator().manual_seed(seed)).cuda().contiguous()
    legal = []
    for _ in range(B):
        m = r.random(A) < (0.6 if seed % 2 else 1.1)
        m[r.integers(0, A)] = True
        legal.append(np.nonzero(m)[0].tolist())
    noises = [r.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal] if seed % 3 else None
    roots = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
    roots.set_tiebreak(0)
    _search_and_replay("ez", model, roots, obs, legal, [-1] * B, noises, S, float(r.choice([0.997, 0.99])), trace=bool(seed % 2))


@pytest.mark.parametrize("seed", range(_OFF, _OFF + 12))
def test_random_two_player_board_search_replays_exactly(seed):
    from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree
    r = np.random.default_rng(1300 + seed)
    gh, gw = [(6, 6), (6, 7), (9, 9), (3, 3), (8, 8), (4, 4)][seed % 6]
    C = int(r.integers(1, 18))
    A = gh * gw + int(r.integers(0, 2))
    B, S = int(r.integers(1, 90)), int(r.integers(1, 80))
    kw = dict(observation_shape=(C, gh, gw), downsample=False, num_res_blocks=int(r.integers(1, 3)))
    if (gh, gw) == (3, 3):
        kw.update(num_channels=16, reward_head_hidden_channels=[8], value_head_hidden_channels=[8], policy_head_hidden_channels=[8],
                  reward_support_range=(-10., 11., 1.), value_support_range=(-10., 11., 1.))
    ez = seed >= 6 and (gh, gw) != (3, 3)      # the second half of the seeds: EfficientZero models on the same boards
    model = (_ez_model if ez else _mz_model)(A, seed=seed, **kw)
    obs = (torch.rand(B, C, gh, gw, generator=torch.Generator().manual_seed(seed)) < 0.4).float().cuda().contiguous()
    legal = []
    for _ in range(B):
        m = r.random(A) < 0.6
        m[A - 1] = True
        legal.append(np.nonzero(m)[0].tolist())
    to_play = r.integers(1, 3, size=B).tolist()
    noises = [r.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal] if seed % 2 else None
    if ez:
        from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree as tree
    else:
        tree = mz_tree
    roots = tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
    roots.set_tiebreak(0)
    _search_and_replay("ez" if ez else "mz", model, roots, obs, legal, to_play, noises, S, 1.0, trace=True)


@pytest.mark.parametrize("seed", range(_OFF, _OFF + 8))
def test_random_mlp_model_search_replays_exactly(seed):
    """vector-observation models (MuZeroModelMLP / EfficientZeroModelMLP): random observation widths, latent widths, action counts"""
    from oracle import torch_models as tm
    r = np.random.default_rng(1400 + seed)
    A, obs_dim = int(r.integers(2, 12)), int(r.integers(2, 30))
    B, S = int(r.integers(1, 200)), int(r.integers(1, 60))
    latent = int(r.choice([128, 256]))
    if seed % 2:
        from lightzero_amd.model.efficientzero_model_mlp import EfficientZeroModelMLP as M
        from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree as tree
        kw = dict(observation_shape=obs_dim, action_space_size=A, lstm_hidden_size=int(r.choice([128, 256])), latent_state_dim=latent,
                  res_connection_in_dynamics=bool(r.integers(0, 2)))
        ref, variant = tm.EfficientZeroModelMLP(**kw), "ez"
    else:
        from lightzero_amd.model.muzero_model_mlp import MuZeroModelMLP as M
        from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree as tree
        kw = dict(observation_shape=obs_dim, action_space_size=A, latent_state_dim=latent)
        ref, variant = tm.MuZeroModelMLP(**kw), "mz"
    from lightzero_amd import _lib as L
    try:
        model = M(**kw).load_state_dict(tm.synthetic_init(ref, seed=seed).state_dict())
    except (L.LzError, ValueError, NotImplementedError) as e:
        assert len(str(e)) > 20
        pytest.skip("refused by the engine: %s" % e)
    obs = torch.randn(B, obs_dim, generator=torch.Generator().manual_seed(seed)).cuda().contiguous()
    legal = []
    for _ in range(B):
        m = r.random(A) < 0.7
        m[r.integers(0, A)] = True
        legal.append(np.nonzero(m)[0].tolist())
    noises = [r.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal]
    roots = tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
    roots.set_tiebreak(0)
    _search_and_replay(variant, model, roots, obs, legal, [-1] * B, noises, S, 0.997, trace=True)


@pytest.mark.parametrize("seed", range(_OFF, _OFF + 3))
def test_tictactoe_efficientzero_two_player_replays_exactly(seed):
    """the reference's TicTacToe EfficientZero configuration (16-channel model, value-prefix LSTM on 16 x 9 + 512 inputs), two players"""
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    r = np.random.default_rng(1500 + seed)
    A, B, S = 9, int(r.integers(1, 70)), int(r.integers(5, 60))
    kw = dict(observation_shape=(3, 3, 3), downsample=False, num_channels=16, reward_head_hidden_channels=[8], value_head_hidden_channels=[8],
              policy_head_hidden_channels=[8], reward_support_range=(-10., 11., 1.), value_support_range=(-10., 11., 1.))
    model = _ez_model(A, seed=seed, **kw)
    obs = (torch.rand(B, 3, 3, 3, generator=torch.Generator().manual_seed(seed)) < 0.4).float().cuda().contiguous()
    legal = []
    for _ in range(B):
        m = r.random(A) < 0.6
        m[r.integers(0, A)] = True
        legal.append(np.nonzero(m)[0].tolist())
    to_play = r.integers(1, 3, size=B).tolist()
    noises = [r.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal]
    roots = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
    roots.set_tiebreak(0)
    _search_and_replay("ez", model, roots, obs, legal, to_play, noises, S, 1.0, trace=True)


@pytest.mark.parametrize("family", ["mz", "ez"])
def test_minigrid_sized_models_replay_exactly(family):
    """the reference's MiniGrid configurations: 2835 observation features (the wide first layer, k_dense_wide), 7 actions; MuZero with
    latent 512, EfficientZero with latent 256 + LSTM 256"""
    from oracle import torch_models as tm
    B, A, S, OBS = 64, 7, 30, 2835
    if family == "mz":
        from lightzero_amd.model.muzero_model_mlp import MuZeroModelMLP as M
        from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree as tree
        kw = dict(observation_shape=OBS, action_space_size=A, latent_state_dim=512)
        ref = tm.MuZeroModelMLP(**kw)
    else:
        from lightzero_amd.model.efficientzero_model_mlp import EfficientZeroModelMLP as M
        from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree as tree
        kw = dict(observation_shape=OBS, action_space_size=A, lstm_hidden_size=256, latent_state_dim=256)
        ref = tm.EfficientZeroModelMLP(**kw)
    model = M(**kw).load_state_dict(tm.synthetic_init(ref, seed=3).state_dict())
    obs = torch.randn(B, OBS, generator=torch.Generator().manual_seed(9)).cuda().contiguous()
    r = np.random.default_rng(2)
    legal = [list(range(A))] * B
    noises = [r.dirichlet([0.3] * A).astype(np.float32).tolist() for _ in range(B)]
    roots = tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
    roots.set_tiebreak(0)
    _search_and_replay(family, model, roots, obs, legal, [-1] * B, noises, S, 0.997, trace=True)


@pytest.mark.parametrize("seed", range(_OFF, _OFF + 8))
def test_random_sampled_efficientzero_search_replays_exactly(seed):
    """Sampled EfficientZero fused loop, device-side draws (the production path) read back per node and injected into the oracle:
    continuous (D = 1..3) and discrete (K of A without replacement) action spaces, random K, batch, simulations, observation width"""
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.tree_search.mcts_ctree import SampledEfficientZeroMCTSCtree
    from test_exact_replay_families_gpu import _sampled_replay, _sez_model
    r = np.random.default_rng(1600 + seed)
    continuous = bool(seed % 2)
    B, S, obs_dim = int(r.integers(1, 150)), int(r.integers(2, 60)), int(r.integers(2, 24))
    if continuous:
        D, K = int(r.integers(1, 4)), int(r.integers(2, 24))
        A = D
    else:
        A = int(r.integers(3, 14))
        D, K = 1, int(r.integers(2, A + 1))
    model = _sez_model(continuous, A, K, obs_dim, seed=50 + seed)
    cfg = dict(num_simulations=S, pb_c_base=19652, pb_c_init=1.25, discount_factor=0.997, value_delta_max=0.01, lstm_horizon_len=5,
               root_noise_weight=0.25, model=dict(action_space_size=A, num_of_sampled_actions=K, continuous_action_space=continuous))
    mcts = SampledEfficientZeroMCTSCtree(cfg)
    legal = [[-1] * K] * B if continuous else [list(range(A))] * B
    roots = mcts.roots(B, legal, A, K, continuous, max_simulations=S)
    roots.set_tiebreak(0, seed=77 + seed)
    obs = torch.randn(B, obs_dim, generator=torch.Generator().manual_seed(seed)).cuda().contiguous()
    noises = r.dirichlet([0.3] * K, size=B).astype(np.float32)
    out = model.initial_inference(obs, roots)
    L.check(L.lib().lz_roots_enable_trace(roots._h, 1))
    roots.prepare_from_inference(0.25, noises, [-1] * B)
    mcts.search(roots, model, out.latent_state, out.reward_hidden_state, [-1] * B)
    node_actions = [roots.get_node_actions(e) for e in range(S + 1)]
    _sampled_replay(model, roots, S, lambda e: node_actions[e], noises, [-1] * B, continuous, A_disc=A)


@pytest.mark.parametrize("seed", range(_OFF, _OFF + 8))
def test_random_gumbel_search_replays_exactly(seed):
    """lz_gsearch (Gumbel MuZero) on conv and vector-observation MuZero models: random action counts, considered-action counts m,
    simulation budgets, ragged legal masks, noise on / off -- records, visit counts, root values, improved policies, completed Q-values"""
    import gumbel_driver as gd
    from oracle import ctree as octree, torch_models as tm
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.tree_search.mcts_ctree import GumbelMuZeroMCTSCtree
    from test_exact_replay_families_gpu import _sims
    r = np.random.default_rng(1700 + seed)
    B, S = int(r.integers(1, 100)), int(r.integers(1, 64))
    if seed % 2:
        from lightzero_amd.model.muzero_model_mlp import MuZeroModelMLP
        A, obs_dim = int(r.integers(2, 20)), int(r.integers(2, 30))
        kw = dict(observation_shape=obs_dim, action_space_size=A, latent_state_dim=int(r.choice([128, 256])))
        model = MuZeroModelMLP(**kw).load_state_dict(tm.synthetic_init(tm.MuZeroModelMLP(**kw), seed=seed).state_dict())
        obs = torch.randn(B, obs_dim, generator=torch.Generator().manual_seed(seed)).cuda().contiguous()
    else:
        from lightzero_amd.model.muzero_model import MuZeroModel
        A = int(r.integers(2, 19))
        model = MuZeroModel(action_space_size=A).load_state_dict(tm.synthetic_init(tm.MuZeroModel(action_space_size=A), seed=seed).state_dict())
        obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(seed)).cuda().contiguous()
    m = int(r.integers(1, min(A, 16) + 1))
    mcts = GumbelMuZeroMCTSCtree(dict(num_simulations=S, discount_factor=0.997, max_num_considered_actions=m, value_delta_max=0.01, root_noise_weight=0.25))
    legal = []
    for _ in range(B):
        k = r.random(A) < 0.7
        k[r.integers(0, A)] = True
        legal.append(np.nonzero(k)[0].tolist())
    noises = [r.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal] if seed % 3 else None
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
    c = dict(B=B, A=A, S=S, m=m, discount=0.997, delta=0.01, noise_w=0.25, legal_list=legal, root_logits=pol0, root_reward=np.zeros(B, np.float32),
             root_value=pred, noises=noises, sims=[dict(r=x["vp"], v=x["v"], logits=x["logits"]) for x in sims])
    o = gd.run_tree(octree.gmz_tree, c, roots_kwargs=dict(action_space_size=A, max_simulations=S))
    dist = np.full((B, A), -1, np.int32)
    for i, d in enumerate(roots.get_distributions()):
        dist[i, :len(d)] = d
    assert (o["distributions"] == dist).all(), "visit counts differ"
    for k, got in (("values", roots.get_values()), ("policies", roots.get_policies(0.997, A)), ("children_values", roots.get_children_values(0.997, A))):
        assert np.array_equal(o[k].view(np.uint32), np.asarray(got, np.float32).view(np.uint32)), "%s not bit-equal" % k


@pytest.mark.parametrize("seed", range(_OFF, _OFF + 8))
def test_random_rezero_search_with_reuse_replays_exactly(seed):
    """lz_search_with_reuse (ReZero): EfficientZero Atari models and two-player MuZero board models, random shapes, ragged legal lists,
    random true actions / reuse values (roots that select their true action skip the network for that simulation)"""
    from lightzero_amd.mcts.tree_search.mcts_ctree import EfficientZeroMCTSCtree, MuZeroMCTSCtree
    from test_exact_replay_families_gpu import _reuse_replay
    r = np.random.default_rng(1800 + seed)
    B, S = int(r.integers(1, 120)), int(r.integers(2, 60))
    if seed % 2:
        gh, gw = [(6, 6), (6, 7), (9, 9), (8, 8)][(seed // 2) % 4]
        C, A = int(r.integers(1, 18)), gh * gw + 1
        model = _mz_model(A, seed=seed, observation_shape=(C, gh, gw), downsample=False)
        mcts = MuZeroMCTSCtree(dict(num_simulations=S, pb_c_base=19652, pb_c_init=1.25, discount_factor=1.0, value_delta_max=0.01, env_type="board_games"))
        obs = (torch.rand(B, C, gh, gw, generator=torch.Generator().manual_seed(seed)) < 0.3).float().cuda().contiguous()
        to_play, variant, disc = r.integers(1, 3, size=B).tolist(), "mz", 1.0
    else:
        A = int(r.integers(2, 19))
        model = _ez_model(A, seed=seed)
        mcts = EfficientZeroMCTSCtree(dict(num_simulations=S, pb_c_base=19652, pb_c_init=1.25, discount_factor=0.997, value_delta_max=0.01, lstm_horizon_len=5))
        obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(seed)).cuda().contiguous()
        to_play, variant, disc = [-1] * B, "ez", 0.997
    legal = []
    for _ in range(B):
        k = r.random(A) < 0.7
        k[A - 1] = True
        legal.append(np.nonzero(k)[0].tolist())
    noises = [r.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal]
    roots = mcts.roots(B, legal, action_space_size=A, max_simulations=S)
    roots.set_tiebreak(0)
    true_action = [int(l[r.integers(0, len(l))]) for l in legal]
    reuse_value = r.standard_normal(B).astype(np.float32).tolist()
    _reuse_replay(variant, model, roots, mcts, obs, legal, to_play, noises, S, disc, true_action, reuse_value)


@pytest.mark.parametrize("seed", range(_OFF, _OFF + 8))
def test_random_conv_sampled_efficientzero_search_replays_exactly(seed):
    """the convolutional Sampled EfficientZero (discrete actions on pixels, round 4): random batch / action / sample counts, both
    observation shapes, GELU and ReLU dynamics, head widths 32..256, 1..3 residual blocks; the device's own draws replayed through the
    oracle sampled tree (tests/test_exact_replay_families_gpu.py::test_conv_sampled_efficientzero_atari_config_replays_exactly)"""
    from oracle import torch_models as tm
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.tree_search.mcts_ctree import SampledEfficientZeroMCTSCtree
    from lightzero_amd.model.sampled_efficientzero_model import SampledEfficientZeroModel
    from test_exact_replay_families_gpu import _sampled_replay
    r = np.random.default_rng(1900 + seed)
    B, S, A = int(r.integers(1, 200)), int(r.integers(1, 50)), int(r.integers(2, 19))
    K = int(r.integers(1, A + 1))
    hw, hid = int(r.choice([64, 96])), int(r.choice([32, 64, 128, 256]))
    kw = dict(observation_shape=(4, hw, hw), action_space_size=A, num_of_sampled_actions=K, downsample=True, continuous_action_space=False,
              norm_type='BN', num_res_blocks=int(r.integers(1, 4)), reward_head_hidden_channels=[hid], value_head_hidden_channels=[hid],
              policy_head_hidden_channels=[hid])
    if seed % 2:
        kw["activation"] = "relu"
    sup = [(-300., 301., 1.), (-50., 51., 1.)][int(r.integers(0, 2))]
    kw.update(reward_support_range=sup, value_support_range=sup)
    ref = tm.synthetic_init(tm.SampledEfficientZeroModel(**kw), seed=seed)
    model = SampledEfficientZeroModel(**kw).load_state_dict(ref.state_dict())
    cfg = dict(num_simulations=S, pb_c_base=19652, pb_c_init=1.25, discount_factor=0.997, value_delta_max=0.01, lstm_horizon_len=5, root_noise_weight=0.25,
               model=dict(action_space_size=A, num_of_sampled_actions=K, continuous_action_space=False))
    mcts = SampledEfficientZeroMCTSCtree(cfg)
    roots = mcts.roots(B, [list(range(A))] * B, A, K, False, max_simulations=S)
    roots.set_tiebreak(0, seed=5 + seed)
    obs = torch.rand(B, 4, hw, hw, generator=torch.Generator().manual_seed(seed)).cuda().contiguous()
    noises = r.dirichlet([0.3] * K, size=B).astype(np.float32)
    out = model.initial_inference(obs, roots)
    L.check(L.lib().lz_roots_enable_trace(roots._h, 1))
    roots.prepare_from_inference(0.25, noises, [-1] * B)
    mcts.search(roots, model, out.latent_state, out.reward_hidden_state, [-1] * B)
    node_actions = [roots.get_node_actions(e) for e in range(S + 1)]
    assert all(np.isfinite(a).all() for a in node_actions)
    ora, _ = _sampled_replay(model, roots, S, lambda e: node_actions[e], noises, [-1] * B, False, A_disc=A)
    assert np.array_equal(ora["root_actions"].view(np.uint32), node_actions[0].view(np.uint32))
    assert (np.asarray(roots.get_distributions()).sum(1) == S).all()


@pytest.mark.parametrize("B,S", [(3001, 12), (1, 120)])
def test_batch_extremes_replay_exactly(B, S):
    """far more roots than any shipped configuration (3,001: twelve workgroups per CU, a ragged last LSTM / head tile), and one root searched deep"""
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    r = np.random.default_rng(77)
    A = 6
    model = _ez_model(A, seed=3)
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(5)).cuda().contiguous()
    legal = [list(range(A))] * B
    noises = [r.dirichlet([0.3] * A).astype(np.float32).tolist() for _ in range(B)]
    roots = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
    roots.set_tiebreak(0)
    _search_and_replay("ez", model, roots, obs, legal, [-1] * B, noises, S, 0.997, trace=False)
