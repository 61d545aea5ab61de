"""the env-step rows the device writes after a search (lz_roots_collect_rows: select_action + packing in one
kernel) against the host packer on the policy's own output dict (lightzero_amd.shard.pack_rows, itself checked against the
reference's GameSegment in tests/test_segment_rows_cpu.py), and the vectorised collect forward against the dict-returning one."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = dict(num_simulations=16, pb_c_base=19652, pb_c_init=1.25, discount_factor=0.997, value_delta_max=0.01,
           lstm_horizon_len=5, root_noise_weight=0.25, root_dirichlet_alpha=0.3)


def _model(A=6, seed=0):
    from oracle import torch_models as tm
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    sd = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A), seed=seed).state_dict()
    return EfficientZeroModel(action_space_size=A).load_state_dict(sd)


def test_device_rows_equal_host_packer():
    from lightzero_amd import _lib as L, shard
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    B, A, S, F = 37, 6, CFG["num_simulations"], 96 * 96
    model = _model(A)
    rng = np.random.default_rng(2)
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(4)).cuda().contiguous()
    mask = (rng.random((B, A)) < 0.6).astype(np.float32)
    mask[np.arange(B), rng.integers(0, A, size=B)] = 1
    legal = [np.nonzero(mask[i])[0].tolist() for i in range(B)]
    to_play = rng.integers(1, 3, size=B).tolist()
    roots = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
    roots.set_tiebreak(0)
    model.initial_inference(obs, roots, fetch=False)
    roots.prepare_from_inference_no_noise(to_play)
    L.check(L.lib().lz_search(roots._h, S, CFG["pb_c_base"], CFG["pb_c_init"], CFG["discount_factor"], CFG["lstm_horizon_len"], CFG["value_delta_max"]))
    for deterministic, T in ((True, 1.0), (False, 0.5)):
        seed = 12345
        dist, cnt, val, pred, logits, pos, ent = roots.get_search_results(select=(T, deterministic, seed))
        out = {i: dict(action=legal[i][pos[i]], visit_count_distributions=dist[i, :cnt[i]].tolist(), visit_count_distribution_entropy=ent[i],
                       searched_value=val[i], predicted_value=pred[i]) for i in range(B)}
        frames = obs[:, -1].reshape(B, -1).cpu().numpy()
        want = shard.pack_rows(out, mask, to_play, A, frames=frames, timestep=list(range(B)))
        W = shard.row_width(A, F)
        rows = torch.full((B, W + 3), -7.0, device="cuda")  # a wider stride than needed: the tail must stay untouched
        hdr, lg = roots.collect_rows(T, deterministic, rows.data_ptr(), W + 3, F, timestep=list(range(B)), seed=seed)
        got = rows.cpu().numpy()
        assert (got[:, W:] == -7.0).all()
        assert np.array_equal(got[:, :shard.HEADER + 2 * A], hdr)
        assert np.array_equal(lg, logits)
        # integers, masks, frames, values: identical; entropy float64 -> float32; child visits: float32 division on both sides
        for col in (shard.F_ACTION, shard.F_REWARD, shard.F_ROOT_VALUE, shard.F_PRED_VALUE, shard.F_TO_PLAY, shard.F_TIMESTEP, shard.F_N_LEGAL):
            assert np.array_equal(got[:, col], want[:, col]), col
        assert np.array_equal(got[:, shard.HEADER + A:W], want[:, shard.HEADER + A:])          # mask + frame
        assert np.array_equal(got[:, shard.HEADER:shard.HEADER + A], want[:, shard.HEADER:shard.HEADER + A])  # child visits
        assert np.allclose(got[:, shard.F_ENTROPY], want[:, shard.F_ENTROPY], rtol=1e-6, atol=1e-7)
        assert all(mask[i, int(got[i, shard.F_ACTION])] == 1 for i in range(B))


def test_forward_collect_rows_matches_dict_forward_and_feeds_the_segment_batch():
    """same search through both collect forwards (deterministic tie-break, eps-greedy arg-max selection, pinned seeds); the rows
    path then drives GameSegmentBatch for a few steps without any per-env loop"""
    from lightzero_amd import shard
    from lightzero_amd.mcts.buffer.game_segment import GameSegmentBatch
    from lightzero_amd.policy.efficientzero import EfficientZeroPolicy
    B, A, F = 32, 6, 96 * 96
    model = _model(A, seed=1)
    cfg = dict(CFG, mcts_tiebreak="first", device_select_action=True, eps=dict(eps_greedy_exploration_in_collect=True))
    pol_a, pol_b = EfficientZeroPolicy(cfg, model), EfficientZeroPolicy(cfg, model)
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(6)).cuda().contiguous()
    rng = np.random.default_rng(3)
    mask = np.ones((B, A), np.float32)  # equal legal counts: both forwards draw the Dirichlet noise with the same np.random call
    rows = torch.zeros(B, shard.row_width(A, F), device="cuda")
    batch = GameSegmentBatch(B, A, 4, (1, 96, 96), frame_stack_num=4)
    batch.reset(obs.cpu().numpy().reshape(B, 4, 1, 96, 96))
    for t in range(3):
        np.random.seed(100 + t)
        out = pol_a._forward_collect(obs, action_mask=mask, temperature=1.0, to_play=[-1] * B, epsilon=0.0)
        np.random.seed(100 + t)
        hdr = pol_b.forward_collect_rows(obs, mask, rows, temperature=1.0, to_play=[-1] * B, timestep=[t] * B, epsilon=0.0)
        assert [int(a) for a in hdr[:, shard.F_ACTION]] == [int(out[i]["action"]) for i in range(B)]
        assert np.array_equal(hdr[:, shard.F_ROOT_VALUE], np.array([out[i]["searched_value"] for i in range(B)], np.float32))
        for i in range(B):
            d = np.asarray(out[i]["visit_count_distributions"], np.float32)
            assert np.array_equal(hdr[i, shard.HEADER:shard.HEADER + len(d)], d / d.sum())
        batch.store_search_stats_rows(hdr)
        batch.append(rows[:, shard.HEADER + 2 * A:].cpu().numpy(), np.zeros(B, np.float32))
    seg = batch.to_arrays(5)
    assert seg["action_segment"].shape == (3,) and seg["obs_segment"].shape == (4 + 3, 1, 96, 96)
    assert np.array_equal(seg["action_mask_segment"][0], mask[5])
    # ragged masks: one gamma draw for the whole batch; every action legal, every row normalised
    ragged = (rng.random((B, A)) < 0.5).astype(np.float32)
    ragged[np.arange(B), rng.integers(0, A, size=B)] = 1
    hdr = pol_b.forward_collect_rows(obs, ragged, rows, temperature=1.0, to_play=[-1] * B)
    assert all(ragged[i, int(hdr[i, shard.F_ACTION])] == 1 for i in range(B))
    assert np.array_equal(hdr[:, shard.HEADER + A:shard.HEADER + 2 * A], ragged)
    assert np.allclose(hdr[:, shard.HEADER:shard.HEADER + A].sum(1), 1.0, atol=1e-6)
    assert np.array_equal(hdr[:, shard.F_N_LEGAL], ragged.sum(1))


def test_device_rows_of_the_sampled_and_gumbel_families_equal_host_packer():
    """lz_roots_collect_rows_ex -- the extra block (root_sampled_actions / improved_policy_probs, game_segment.py:254-258)
    written on the device == shard.pack_rows on the policy-style output dict (itself == the reference GameSegment,
    tests/test_segment_rollover_cpu.py), and the rows feed GameSegmentBatch"""
    from oracle import torch_models as tm
    from lightzero_amd import _lib as L, shard
    from lightzero_amd.mcts.buffer.game_segment import GameSegmentBatch
    from lightzero_amd.mcts.tree_search.mcts_ctree import SampledEfficientZeroMCTSCtree, GumbelMuZeroMCTSCtree
    from lightzero_amd.model.sampled_efficientzero_model_mlp import SampledEfficientZeroModelMLP
    from lightzero_amd.model.muzero_model import MuZeroModel
    from lightzero_amd.policy.utils import select_action
    # ---- Sampled EfficientZero, continuous actions (BASELINE configs[4] shape)
    B, D, K, S = 48, 2, 20, 24
    ref = tm.synthetic_init(tm.SampledEfficientZeroModelMLP(observation_shape=5, action_space_size=D, continuous_action_space=True, num_of_sampled_actions=K), seed=3)
    model = SampledEfficientZeroModelMLP(observation_shape=5, action_space_size=D, continuous_action_space=True, num_of_sampled_actions=K).load_state_dict(ref.state_dict())
    cfg = dict(CFG, num_simulations=S, model=dict(action_space_size=D, num_of_sampled_actions=K, continuous_action_space=True))
    mcts = SampledEfficientZeroMCTSCtree(cfg)
    roots = mcts.roots(B, [[-1] * K] * B, D, K, True, max_simulations=S)
    roots.set_tiebreak(0, seed=5)
    obs = torch.randn(B, 5, generator=torch.Generator().manual_seed(1)).cuda().contiguous()
    out = model.initial_inference(obs, roots)
    roots.prepare_from_inference(0.25, None, [-1] * B)
    mcts.search(roots, model, out.latent_state, out.reward_hidden_state, [-1] * B)
    dist, val, acts = roots.get_distributions(), roots.get_values(), np.asarray(roots.get_sampled_actions(), np.float32)
    F, E = 5, K * D
    W = shard.row_width(K, F, E)
    rows = torch.zeros(B, W, device="cuda")
    torch.cuda.synchronize()   # raw-pointer call on the engine's own stream: torch's zero-fill must have landed first
    hdr, pol = roots.collect_rows(1.0, True, rows.data_ptr(), W, F, timestep=list(range(B)), seed=9)
    got = rows.cpu().numpy()
    assert np.array_equal(got[:, :shard.HEADER + 2 * K + E], hdr) and np.array_equal(pol, out.policy_logits)
    o = {}
    for i in range(B):
        idx, ent = select_action(dist[i], temperature=1.0, deterministic=True)
        o[i] = dict(action=acts[i][idx], visit_count_distributions=dist[i], visit_count_distribution_entropy=ent, searched_value=val[i],
                    predicted_value=out.value[i], root_sampled_actions=acts[i])
    want = shard.pack_rows(o, np.ones((B, K), np.float32), [-1] * B, K, frames=obs.cpu().numpy(), timestep=list(range(B)), extra_key="root_sampled_actions")
    # (duplicated sampled actions share a child: the arg-max position is the first of the class on both sides)
    for col in (shard.F_ACTION, shard.F_ROOT_VALUE, shard.F_PRED_VALUE, shard.F_TO_PLAY, shard.F_TIMESTEP, shard.F_N_LEGAL):
        assert np.array_equal(got[:, col], want[:, col]), col
    assert np.array_equal(got[:, shard.HEADER:], want[:, shard.HEADER:])   # visits, mask, sampled actions, frame
    assert np.allclose(got[:, shard.F_ENTROPY], want[:, shard.F_ENTROPY], rtol=1e-6, atol=1e-7)
    cols = shard.unpack_rows(got, K, extra_words=E)
    assert np.array_equal(cols["extra"].reshape(B, K, D), acts)
    seg = GameSegmentBatch(B, K, 4, (5,), sampled_actions_shape=(K, D))
    seg.reset(obs.cpu().numpy().reshape(B, 1, 5))
    seg.store_search_stats_rows(hdr)
    seg.append(obs.cpu().numpy(), np.zeros(B))
    assert np.array_equal(seg.to_arrays(3)["root_sampled_actions"][0], acts[3])
    # ---- Gumbel MuZero
    B, A, S, m = 40, 6, 20, 4
    refm = tm.synthetic_init(tm.MuZeroModel(action_space_size=A), seed=4)
    gmodel = MuZeroModel(action_space_size=A).load_state_dict(refm.state_dict())
    gcfg = dict(num_simulations=S, discount_factor=0.997, max_num_considered_actions=m, value_delta_max=0.01, root_noise_weight=0.25)
    g = GumbelMuZeroMCTSCtree(gcfg)
    rng = np.random.default_rng(6)
    mask = (rng.random((B, A)) < 0.7).astype(np.float32); mask[:, 3] = 1
    legal = [np.nonzero(x)[0].tolist() for x in mask]
    groots = g.roots(B, legal, action_space_size=A, max_simulations=S)
    gobs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(2)).cuda().contiguous()
    gout = gmodel.initial_inference(gobs, groots)
    groots.prepare_from_inference_no_noise([-1] * B)
    g.search(groots, gmodel, gout.latent_state, [-1] * B)
    gd, gv = groots.get_distributions(), groots.get_values()
    improved = np.asarray(groots.get_policies(0.997, A), np.float32)
    F = 96 * 96
    W = shard.row_width(A, F, A)
    grows = torch.zeros(B, W, device="cuda")
    torch.cuda.synchronize()   # raw-pointer call on the engine's own stream: torch's zero-fill must have landed first
    ghdr, _ = groots.collect_rows(1.0, True, grows.data_ptr(), W, F, discount=0.997)
    ggot = grows.cpu().numpy()
    o = {}
    for i in range(B):
        _, ent = select_action(gd[i], temperature=1.0, deterministic=True)
        o[i] = dict(action=int(np.argmax(np.where(mask[i] == 1.0, improved[i], 0.0))), visit_count_distributions=gd[i],
                    visit_count_distribution_entropy=ent, searched_value=gv[i], predicted_value=gout.value[i], improved_policy_probs=improved[i])
    gwant = shard.pack_rows(o, mask, [-1] * B, A, frames=gobs[:, -1].reshape(B, -1).cpu().numpy(), extra_key="improved_policy_probs")
    for col in (shard.F_ACTION, shard.F_ROOT_VALUE, shard.F_PRED_VALUE, shard.F_TO_PLAY, shard.F_TIMESTEP, shard.F_N_LEGAL):
        assert np.array_equal(ggot[:, col], gwant[:, col]), col
    assert np.array_equal(ggot[:, shard.HEADER:], gwant[:, shard.HEADER:])
    assert np.array_equal(ggot[:, :shard.HEADER + 3 * A], ghdr)


def test_forward_collect_rows_with_pure_policy():
    """``collect_with_pure_policy`` on the rows path (efficientzero.py:597,644-655; muzero_collector.py:98-99,505,596): no search; the
    action is drawn from softmax(policy logits over the legal actions) -- the SAME draw distribution as the dict-returning forward --,
    searched value = predicted value, the stored visit statistics are the collector's zero list; the device rows carry the header and
    the newest frame."""
    from lightzero_amd import shard
    from lightzero_amd.policy.efficientzero import EfficientZeroPolicy
    B, A, F = 64, 6, 96 * 96
    model = _model(A, seed=2)
    pol = EfficientZeroPolicy(dict(CFG, collect_with_pure_policy=True), model)
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(8)).cuda().contiguous()
    rng = np.random.default_rng(5)
    mask = (rng.random((B, A)) < 0.6).astype(np.float32)
    mask[np.arange(B), rng.integers(0, A, size=B)] = 1
    rows = torch.zeros(B, shard.row_width(A, F), device="cuda")
    np.random.seed(3)
    hdr = pol.forward_collect_rows(obs, mask, rows, temperature=1.0, to_play=[-1] * B, timestep=list(range(B)))
    out = pol._forward_collect(obs, action_mask=mask, temperature=1.0, to_play=[-1] * B)
    HW = shard.HEADER + 2 * A
    got = rows.cpu().numpy()
    assert np.array_equal(got[:, :HW], hdr)
    assert np.array_equal(got[:, HW:], obs[:, 3].reshape(B, -1).cpu().numpy())          # the newest frame
    acts = hdr[:, shard.F_ACTION].astype(int)
    assert all(mask[i, acts[i]] == 1 for i in range(B))
    pred = np.array([float(np.asarray(out[i]["predicted_value"]).reshape(-1)[0]) for i in range(B)], np.float32)
    assert np.array_equal(hdr[:, shard.F_PRED_VALUE], pred) and np.array_equal(hdr[:, shard.F_ROOT_VALUE], pred)
    assert (hdr[:, shard.HEADER:shard.HEADER + A] == 0).all() and (hdr[:, shard.F_ENTROPY] == 0).all()
    assert np.array_equal(hdr[:, shard.HEADER + A:HW], mask) and np.array_equal(hdr[:, shard.F_N_LEGAL], mask.sum(1))
    assert np.array_equal(hdr[:, shard.F_TIMESTEP], np.arange(B, dtype=np.float32))
    # the draw follows softmax(logits over the legal actions): 400 draws of one env against its probabilities
    logits = np.array(out[0]["predicted_policy_logits"], np.float64)
    z = np.where(mask[0] != 0, logits, -np.inf)
    p = np.exp(z - z.max()); p /= p.sum()
    one = obs[:1].repeat(B, 1, 1, 1).contiguous()
    m1 = np.repeat(mask[:1], B, 0)
    cnt = np.zeros(A)
    for k in range(7):
        h = pol.forward_collect_rows(one, m1, rows, temperature=1.0, to_play=[-1] * B)
        cnt += np.bincount(h[:, shard.F_ACTION].astype(int), minlength=A)
    n = cnt.sum()
    assert (cnt[mask[0] == 0] == 0).all()
    assert np.all(np.abs(cnt / n - p) < 4.5 * np.sqrt(p * (1 - p) / n) + 1e-3), (cnt / n, p)
