"""The exact end-to-end gate on the PRODUCTION path: lz_search runs as the benchmark runs it (captured HIP graph, tree step
fused into the chain launch's prologue where the tree fits LDS, no tracing), then the device's OWN per-simulation network outputs
(pool slots 1..S: value prefix | reward, value, policy logits; slot 0: root logits) are replayed through the CPU tree oracle
(oracle/ctree_oracle.c, pinned bit-exact to the reference's compiled ctree) and -- where oracle/_ref is present -- through the
reference's own compiled ctree (rand() -> 0 build).  Required: 100 % of the roots with identical visit-count distributions,
bit-equal root values and bit-equal min-max statistics; with tracing on (one D2D copy per simulation inside the captured graph)
also the identical (parent slot, action, search length, to_play) record of every simulation.

Why this is the exact gate: the tree half is integer / order-sensitive fp32 work that must be bit-exact, the network half is
fp32 with a different summation order than torch (compared at 2e-5 / 3e-4 in test_nn_golden_gpu.py).  Replaying the device's
own network outputs removes the network's rounding from the comparison, so any difference left is a tree, gather or
launch-sequence bug (a wrong latent slot fed to the network shows up as a record mismatch under tracing and as diverging
visit counts otherwise, because the outputs then belong to another (parent, action)).

Sizes: BASELINE configs[1] full size (256 x 50), configs[2] full size (1024 x 400, deep trees: the non-fused tree step),
configs[3] per-GPU share (Go 9x9, 64 x 200, two players, ragged legal masks), reanalyze-shaped batches (1536 roots,
prepare_no_noise; and a two-player Go batch with per-root to_play), MuZero Atari."""
import numpy as np
import pytest
import torch

import tree_driver as td

pytestmark = pytest.mark.gpu

PB = dict(pb_c_base=19652, pb_c_init=1.25, delta=0.01, horizon=5)


def _oracle_mods(variant):
    """[(name, module, roots_kwargs)]: the C restatement always, the reference's compiled ctree when it is on this box"""
    from oracle import build_ref, ctree as octree
    mods = [("oracle/ctree_oracle.c", octree.ez_tree if variant == "ez" else octree.mz_tree, True)]
    ref = build_ref.load("det")
    if ref:
        mods.append(("oracle/_ref/det (the reference's own ctree)", ref[0] if variant == "ez" else ref[1], False))
    return mods


def _search_and_replay(variant, model, roots, obs, legal, to_play, noises, S, discount, noise_w=0.25, trace=False):
    from lightzero_amd import _lib as L
    lib = L.lib()
    B, A = roots.num, model.action_space_size
    out = model.initial_inference(obs, roots, fetch=False)
    assert out is None
    L.check(lib.lz_roots_enable_trace(roots._h, 1 if trace else 0))
    if noises is not None:
        roots.prepare_from_inference(noise_w, noises, to_play)
    else:
        roots.prepare_from_inference_no_noise(to_play)
    L.check(lib.lz_search(roots._h, S, PB["pb_c_base"], PB["pb_c_init"], discount, PB["horizon"] if variant == "ez" else 0, PB["delta"]))
    dist, cnt, val, pred, logits0 = roots.get_search_results()
    d_dist = [dist[i, :cnt[i]].tolist() for i in range(B)]
    d_mm = roots.get_minmax()
    sims = []
    for s in range(1, S + 1):
        vp = np.zeros(B, np.float32); v = np.zeros(B, np.float32); lg = np.zeros((B, A), np.float32)
        L.check(lib.lz_roots_read_sim_outputs(roots._h, s, vp, v, lg.reshape(-1)))
        sims.append(dict(vp=vp, v=v, logits=lg))
    rec = None
    if trace:
        tr = np.zeros((S, B, 4), np.int32)
        L.check(lib.lz_roots_read_trace(roots._h, S, tr.reshape(-1)))
        rec = tr
    case = dict(variant=variant, B=B, A=A, S=S, legal_list=[list(l) for l in legal], to_play_list=list(to_play),
                root_logits=logits0, root_vp=np.zeros(B, np.float32), noises=noises, noise_w=noise_w, sims=sims,
                discount=discount, **PB)
    for name, mod, sized in _oracle_mods(variant):
        o = td.run_tree(mod, case, roots_kwargs=dict(action_space_size=A, max_simulations=S) if sized else None)
        same = sum(int(a == b) for a, b in zip(o["distributions"], d_dist))
        assert same == B, "%s: only %d / %d roots have identical visit-count distributions" % (name, same, B)
        assert np.array_equal(o["values"].view(np.uint32), np.asarray(val, np.float32).view(np.uint32)), "%s: root values not bit-equal" % name
        if "minmax" in o:
            assert np.array_equal(o["minmax"].view(np.uint32), np.asarray(d_mm, np.float32).view(np.uint32)), "%s: min-max stats not bit-equal" % name
        if rec is not None:  # (ix, action, search_len, virtual_to_play) of every simulation
            assert np.array_equal(o["records"][:, :, [0, 2, 3, 4]], rec), "%s: per-simulation selection records differ" % name
    assert all(sum(d) == S for d in d_dist)
    return d_dist, np.asarray(val), sims


def _ez_model(A=6, seed=0, **kw):
    from oracle import torch_models as tm
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    sd = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A, **kw), seed=seed).state_dict()
    return EfficientZeroModel(action_space_size=A, **kw).load_state_dict(sd)


def _mz_model(A=4, seed=0, **kw):
    from oracle import torch_models as tm
    from lightzero_amd.model.muzero_model import MuZeroModel
    sd = tm.synthetic_init(tm.MuZeroModel(action_space_size=A, **kw), seed=seed).state_dict()
    return MuZeroModel(action_space_size=A, **kw).load_state_dict(sd)


def test_configs1_full_size_production_path_replays_exactly():
    """BASELINE configs[1]: EfficientZero Atari 96x96x4, 256 roots x 50 simulations, Dirichlet noise -- first exactly as
    bench.py runs it (graph + fused tree step, no trace), then the same search with tracing captured into the graph."""
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    B, A, S = 256, 6, 50
    model = _ez_model(A)
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(21)).cuda().contiguous()
    rng = np.random.default_rng(3)
    noises = [rng.dirichlet([0.3] * A).astype(np.float32).tolist() for _ in range(B)]
    legal = [list(range(A))] * B
    roots = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
    roots.set_tiebreak(0)
    d1, v1, sims1 = _search_and_replay("ez", model, roots, obs, legal, [-1] * B, noises, S, 0.997)
    roots.reset(legal)
    d2, v2, sims2 = _search_and_replay("ez", model, roots, obs, legal, [-1] * B, noises, S, 0.997, trace=True)
    # tracing must not change the search, and the device is bit-reproducible run to run
    assert d1 == d2 and np.array_equal(v1.view(np.uint32), v2.view(np.uint32))
    assert all(np.array_equal(a["v"], b["v"]) and np.array_equal(a["logits"], b["logits"]) for a, b in zip(sims1, sims2))


def test_configs1_sharp_prior_deep_paths_replay_exactly():
    """VERDICT r4 #1c: the 8d recipe's near-uniform priors only ever build paths of depth <= 3 at configs[1].  The same 256 x 50 search
    with the policy and value heads' last layers x 10 (a trained agent's sharp prior; search paths up to ~10 deep, so the tree step
    spends most of its time below the root) through the same gate -- with the records of every simulation, and with the
    stochastic-tie-break graph's sibling, the level-by-level walk (LZ_TRAVERSE_SERIAL=1), required to give the identical search."""
    import os
    from oracle import torch_models as tm
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    from lightzero_amd.model.synthetic import sharpen_state_dict
    B, A, S = 256, 6, 50
    sd = sharpen_state_dict(tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A), seed=0).state_dict(), 10.0)
    model = EfficientZeroModel(action_space_size=A).load_state_dict(sd)
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(31)).cuda().contiguous()
    rng = np.random.default_rng(13)
    noises = [rng.dirichlet([0.3] * A).astype(np.float32).tolist() for _ in range(B)]
    legal = [list(range(A))] * B
    roots = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
    roots.set_tiebreak(0)
    d1, v1, _ = _search_and_replay("ez", model, roots, obs, legal, [-1] * B, noises, S, 0.997, trace=True)
    from lightzero_amd import _lib as L
    depth = np.zeros((B, S), np.int32)
    L.check(L.lib().lz_roots_get_node_depths(roots._h, S, depth.reshape(-1)))
    tr = np.zeros((S, B, 4), np.int32)
    L.check(L.lib().lz_roots_read_trace(roots._h, S, tr.reshape(-1)))
    assert np.array_equal(depth, tr[:, :, 2].T), "lz_roots_get_node_depths differs from the traced search lengths"
    assert depth.max() >= 6 and depth.mean() > 3.0, "sharp priors did not deepen the search (mean %.2f, max %d)" % (depth.mean(), depth.max())
    os.environ["LZ_TRAVERSE_SERIAL"] = "1"
    try:
        roots.reset(legal)
        d2, v2, _ = _search_and_replay("ez", model, roots, obs, legal, [-1] * B, noises, S, 0.997, trace=True)
    finally:
        os.environ.pop("LZ_TRAVERSE_SERIAL", None)
    assert d1 == d2 and np.array_equal(v1.view(np.uint32), v2.view(np.uint32))


def test_reanalyze_shaped_batch_replays_exactly():
    """game_buffer_efficientzero.py:325-409 -- batch_size * (unroll + 1) = 256 * 6 = 1536 roots, no noise."""
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    B, A, S = 1536, 6, 50
    model = _ez_model(A, seed=2)
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(22)).cuda().contiguous()
    rng = np.random.default_rng(4)
    legal = []
    for _ in range(B):  # ragged legal masks, as stored with the trajectories
        m = rng.random(A) < 0.75
        m[rng.integers(0, A)] = True
        legal.append(np.nonzero(m)[0].tolist())
    roots = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
    roots.set_tiebreak(0)
    _search_and_replay("ez", model, roots, obs, legal, [-1] * B, None, S, 0.997, trace=True)


def test_configs2_deep_tree_muzero_full_size_replays_exactly():
    """BASELINE configs[2]: MuZero Atari, 1024 roots x 400 simulations (trees beyond the LDS budget: the HBM tree step)."""
    from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree
    B, A, S = 1024, 4, 400
    model = _mz_model(A, seed=3)
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(23)).cuda().contiguous()
    rng = np.random.default_rng(5)
    noises = [rng.dirichlet([0.3] * A).astype(np.float32).tolist() for _ in range(B)]
    legal = [list(range(A))] * B
    roots = mz_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
    roots.set_tiebreak(0)
    _search_and_replay("mz", model, roots, obs, legal, [-1] * B, noises, S, 0.997)


def test_muzero_atari_fused_replays_exactly_with_trace():
    """the MuZero instantiation of the fused chain prologue (k_chain<6,6,false,2>) at 256 x 50, records included"""
    from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree
    B, A, S = 256, 4, 50
    model = _mz_model(A, seed=4)
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(24)).cuda().contiguous()
    rng = np.random.default_rng(6)
    noises = [rng.dirichlet([0.3] * A).astype(np.float32).tolist() for _ in range(B)]
    legal = [list(range(A))] * B
    roots = mz_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
    roots.set_tiebreak(0)
    _search_and_replay("mz", model, roots, obs, legal, [-1] * B, noises, S, 0.997, trace=True)


def _go_batch(B, seed):
    rng = np.random.default_rng(seed)
    A = 82
    obs = (torch.rand(B, 17, 9, 9, generator=torch.Generator().manual_seed(seed)) < 0.3).float().cuda().contiguous()
    legal = []
    for _ in range(B):
        m = rng.random(A) < 0.7
        m[A - 1] = True  # pass
        legal.append(np.nonzero(m)[0].tolist())
    to_play = rng.integers(1, 3, size=B).tolist()
    noises = [rng.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal]
    return obs, legal, to_play, noises


def test_configs3_go9_two_player_replays_exactly():
    """BASELINE configs[3], one GPU's share: Go 9x9 MuZero, 64 roots x 200 simulations, A = 82, to_play in {1, 2}."""
    from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree
    B, A, S = 64, 82, 200
    model = _mz_model(A, seed=5, observation_shape=(17, 9, 9), downsample=False)
    obs, legal, to_play, noises = _go_batch(B, 25)
    roots = mz_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
    roots.set_tiebreak(0)
    _search_and_replay("mz", model, roots, obs, legal, to_play, noises, S, 1.0, trace=True)


def test_reanalyze_go_two_player_no_noise_replays_exactly():
    """SURVEY 8(f2), MuZero variant (game_buffer_muzero.py:648-672): a reanalyze batch of board positions with per-root to_play
    and legal masks, prepare_no_noise."""
    from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree
    B, A, S = 384, 82, 50
    model = _mz_model(A, seed=6, observation_shape=(17, 9, 9), downsample=False)
    obs, legal, to_play, _ = _go_batch(B, 26)
    roots = mz_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
    roots.set_tiebreak(0)
    _search_and_replay("mz", model, roots, obs, legal, to_play, None, S, 1.0)


def test_weight_refresh_on_live_roots():
    """search, load_state_dict with other weights on the same engine, search on the SAME roots (their captured
    graph holds weight pointers): must equal a search on fresh roots, and must differ from the first one."""
    from oracle import torch_models as tm
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    B, A, S = 32, 6, 20
    model = _ez_model(A, seed=7)
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(27)).cuda().contiguous()
    legal = [list(range(A))] * B
    roots = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
    roots.set_tiebreak(0)
    d1, v1, _ = _search_and_replay("ez", model, roots, obs, legal, [-1] * B, None, S, 0.997)
    model.load_state_dict(tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A), seed=8).state_dict())
    roots.reset(legal)
    d2, v2, _ = _search_and_replay("ez", model, roots, obs, legal, [-1] * B, None, S, 0.997)
    fresh = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
    fresh.set_tiebreak(0)
    d3, v3, _ = _search_and_replay("ez", model, fresh, obs, legal, [-1] * B, None, S, 0.997)
    assert d2 == d3 and np.array_equal(v2.view(np.uint32), v3.view(np.uint32)), "reused roots ran with stale weights"
    assert not np.array_equal(v1, v2), "the refreshed weights had no effect"


def test_two_models_in_one_process_do_not_share_an_engine():
    """a second model object must not replace the first one's weights"""
    from lightzero_amd import _lib as L
    a = _ez_model(6, seed=9)
    b = _mz_model(4, seed=10)
    assert getattr(a.engine, "value", a.engine) != getattr(b.engine, "value", b.engine)
    a._check_owner(); b._check_owner()
    # and an explicit re-use of an engine is detected on the stale object
    from lightzero_amd.model.muzero_model import MuZeroModel
    MuZeroModel(action_space_size=4, engine=a.engine)
    with pytest.raises(L.LzError):
        a._check_owner()


def test_gomoku_narrow_model_two_player_replays_exactly():
    """the reference's gomoku configuration (32 channels, 6x6 board, A = 36, supports (-10, 11, 1), two players): the narrow chain
    kernel (k_chain_small) in the search loop"""
    from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree
    B, A, S = 32, 36, 50
    kw = dict(observation_shape=(3, 6, 6), downsample=False, num_channels=32, reward_support_range=(-10., 11., 1.), value_support_range=(-10., 11., 1.))
    model = _mz_model(A, seed=11, **kw)
    rng = np.random.default_rng(12)
    obs = (torch.rand(B, 3, 6, 6, generator=torch.Generator().manual_seed(28)) < 0.3).float().cuda().contiguous()
    legal = []
    for _ in range(B):
        m = rng.random(A) < 0.7
        m[rng.integers(0, A)] = True
        legal.append(np.nonzero(m)[0].tolist())
    to_play = rng.integers(1, 3, size=B).tolist()
    noises = [rng.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal]
    roots = mz_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
    roots.set_tiebreak(0)
    _search_and_replay("mz", model, roots, obs, legal, to_play, noises, S, 1.0, trace=True)


def test_two_residual_blocks_replays_exactly():
    """num_res_blocks = 2: nine convolutions in the recurrent chain launch, tree step fused"""
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    B, A, S = 64, 6, 30
    model = _ez_model(A, seed=13, num_res_blocks=2)
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(29)).cuda().contiguous()
    rng = np.random.default_rng(14)
    noises = [rng.dirichlet([0.3] * A).astype(np.float32).tolist() for _ in range(B)]
    legal = [list(range(A))] * B
    roots = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
    roots.set_tiebreak(0)
    _search_and_replay("ez", model, roots, obs, legal, [-1] * B, noises, S, 0.997, trace=True)


@pytest.mark.parametrize("B,A,S,ragged", [(1, 18, 12, False), (33, 3, 20, True), (300, 18, 30, True), (17, 64, 9, True)])
def test_odd_batches_and_action_counts_replay_exactly(B, A, S, ragged):
    """Edge shapes of the fused EfficientZero loop (split heads, tree step in the chain launch): one root; a batch that is neither a
    multiple of the 16-row LSTM / head tiles nor of the CU count; the full 18-action Atari set; 64 actions = one lane per action."""
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    model = _ez_model(A, seed=5)
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(30 + B)).cuda().contiguous()
    rng = np.random.default_rng(B * 131 + A)
    legal = []
    for _ in range(B):
        m = rng.random(A) < 0.7 if ragged else np.ones(A, bool)
        m[rng.integers(0, A)] = True
        legal.append(np.nonzero(m)[0].tolist())
    noises = [rng.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal]
    roots = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
    roots.set_tiebreak(0)
    _search_and_replay("ez", model, roots, obs, legal, [-1] * B, noises, S, 0.997, trace=True)
