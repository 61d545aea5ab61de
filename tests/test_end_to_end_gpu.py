"""GPU end-to-end: the policy surface (EfficientZeroPolicy._forward_collect / _forward_eval) on the
engine vs the oracle pipeline (reference-style driver + torch restatement + CPU ctree oracle)."""
import numpy as np
import pytest
from parity_util import assert_root_values_close
import torch

pytestmark = pytest.mark.gpu

CFG = dict(num_simulations=20, pb_c_base=19652, pb_c_init=1.25, discount_factor=0.997, value_delta_max=0.01,
           lstm_horizon_len=5, root_noise_weight=0.25, root_dirichlet_alpha=0.3)


def test_fused_search_vs_oracle_pipeline():
    """Same obs / weights / noise, deterministic tie-break on both sides.  The trees see network outputs that
    differ by ~1e-6 (and value scalars quantised at ~1.3e-4 by the reference's h^-1), so a few arg-max decisions
    can flip; require >= 90% of roots with IDENTICAL visit distributions and, on those, root values within 2e-3 for >= 95% of
    them (a decision below the root can flip without changing the root's visit counts: parity_util.assert_root_values_close)."""
    from oracle import ctree as octree, search as osearch, torch_models as tm
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    B, A, S = 64, 6, CFG["num_simulations"]
    ref = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A))
    model = EfficientZeroModel(action_space_size=A).load_state_dict(ref.state_dict())
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(5))
    rng = np.random.default_rng(0)
    noises = [rng.dirichlet([0.3] * A).astype(np.float32).tolist() for _ in range(B)]
    legal = [list(range(A))] * B
    rec_o = []
    o_dist, o_val, o_pred, o_logits = osearch.ez_forward_collect(
        octree.ez_tree, ref, obs, legal, noises, [-1] * B, CFG, roots_kwargs=dict(action_space_size=A, max_simulations=S), record=rec_o)
    roots = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S)
    roots.set_tiebreak(0)
    out = model.initial_inference(obs.cuda().contiguous(), roots)
    roots.prepare_from_inference(CFG["root_noise_weight"], noises, [-1] * B)
    L.check(L.lib().lz_roots_enable_trace(roots._h, 1))
    L.check(L.lib().lz_search(roots._h, S, CFG["pb_c_base"], CFG["pb_c_init"], CFG["discount_factor"],
                              CFG["lstm_horizon_len"], CFG["value_delta_max"]))
    d_dist, d_val = roots.get_distributions(), np.array(roots.get_values())
    same = np.array([a == b for a, b in zip(o_dist, d_dist)])
    print("identical visit distributions: %d / %d; max |d root value| on those: %.2e; pred value max diff %.2e" %
          (same.sum(), B, np.abs(np.array(o_val) - d_val)[same].max(), np.abs(o_pred - out.value).max()))
    # recorded + every differing root attributed (tests/e2e_common.py); 64 / 64 measured (profiles/r06_parity.json): one root of margin
    import e2e_common
    import parity_record
    e2e_common.attribute_and_gate("e2e/ez_atari96/B%d_S%d" % (B, S), "ez", octree.ez_tree, CFG, A, legal, noises, [-1] * B, o_logits,
                                  np.asarray(out.policy_logits, np.float32), rec_o, e2e_common.device_records(roots, L.lib(), L, B, A, S),
                                  o_dist, d_dist, o_val, d_val, gate=0.98)
    assert_root_values_close(o_val, d_val, same)
    assert np.abs(o_pred - out.value).max() < 3e-4
    ol = np.asarray(o_logits, np.float64)
    parity_record.check("e2e/ez_atari96/root_policy/B%d" % B, {"policy": float(np.max(np.abs(ol - out.policy_logits) / (1.0 + np.abs(ol))))})


def test_policy_forward_collect_and_eval_contract():
    """efficientzero.py:636-643 output contract; legality like lzero/mcts/tests/test_mcts_ctree.py:272-283."""
    from oracle import torch_models as tm
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    from lightzero_amd.policy.efficientzero import EfficientZeroPolicy
    B, A = 12, 6
    ref = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A))
    model = EfficientZeroModel(action_space_size=A).load_state_dict(ref.state_dict())
    policy = EfficientZeroPolicy(dict(CFG, num_simulations=10), model)
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(7)).cuda()
    rng = np.random.default_rng(1)
    mask = (rng.random((B, A)) < 0.6).astype(np.float32)
    mask[:, 0] = 1
    ids = np.arange(100, 100 + B)
    out = policy._forward_collect(obs, action_mask=mask, temperature=1.0, to_play=[-1] * B, epsilon=0.0, ready_env_id=ids)
    assert sorted(out) == ids.tolist()
    for j, i in enumerate(ids):
        o = out[i]
        assert set(o) == {"action", "visit_count_distributions", "visit_count_distribution_entropy", "searched_value",
                          "predicted_value", "predicted_policy_logits"}
        assert mask[j, o["action"]] == 1
        assert len(o["visit_count_distributions"]) == int(mask[j].sum()) and sum(o["visit_count_distributions"]) == 10
        assert len(o["predicted_policy_logits"]) == A
    ev = policy._forward_eval(obs.cpu().numpy(), action_mask=mask, to_play=[-1] * B, ready_env_id=ids)
    for j, i in enumerate(ids):
        d = ev[i]["visit_count_distributions"]
        legal = np.nonzero(mask[j])[0]
        assert ev[i]["action"] == legal[int(np.argmax(d))]
    det = EfficientZeroPolicy(dict(CFG, num_simulations=10, mcts_tiebreak="first"), model)
    ev = det._forward_eval(obs.cpu().numpy(), action_mask=mask, to_play=[-1] * B, ready_env_id=ids)
    ev2 = det._forward_eval(obs, action_mask=mask, to_play=[-1] * B, ready_env_id=ids)  # no noise + first-arg-max: repeatable
    assert all(ev[i]["visit_count_distributions"] == ev2[i]["visit_count_distributions"] for i in ids)


def test_foreign_torch_model_uses_device_tree():
    """Plumbing path: the reference loop with a torch model, tree kernels on the device; must agree with the
    same loop over the CPU oracle tree (identical network outputs => identical trees)."""
    from oracle import ctree as octree, search as osearch, torch_models as tm
    from lightzero_amd.mcts.tree_search.mcts_ctree import EfficientZeroMCTSCtree
    B, A, S = 8, 6, 10
    ref = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A))
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(9))
    cfg = dict(CFG, num_simulations=S, device="cpu")
    with torch.no_grad():
        o = ref.initial_inference(obs)
    lat = o.latent_state.numpy(); rh = (o.reward_hidden_state[0].numpy(), o.reward_hidden_state[1].numpy())
    logits = o.policy_logits.numpy().tolist()
    legal = [list(range(A))] * B
    mcts = EfficientZeroMCTSCtree(cfg)
    roots = mcts.roots(B, legal, action_space_size=A, max_simulations=S)
    roots.set_tiebreak(0)
    roots.prepare_no_noise([0.] * B, logits, [-1] * B)
    mcts.search(roots, ref, lat, rh, [-1] * B)
    oroots = octree.ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S)
    oroots.prepare_no_noise([0.] * B, logits, [-1] * B)
    osearch.ez_search(octree.ez_tree, oroots, ref, lat, rh, [-1] * B, cfg)
    assert roots.get_distributions() == oroots.get_distributions()
    assert np.allclose(roots.get_values(), oroots.get_values(), atol=1e-6)


@pytest.mark.parametrize("tiebreak", [0, 1])
def test_lds_staged_tree_step_is_bit_identical_to_the_hbm_one(tiebreak):
    """k_backprop_traverse_lds (tree of one root staged in LDS, stores written through) vs k_backprop_traverse (HBM):
    same search, same network outputs -> identical per-simulation records, visit counts, root values (bitwise) and
    min-max statistics."""
    import os
    from oracle import torch_models as tm
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    B, A, S = 48, 6, 50
    ref = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A))
    model = EfficientZeroModel(action_space_size=A).load_state_dict(ref.state_dict())
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(11)).cuda().contiguous()
    rng = np.random.default_rng(3)
    mask = (rng.random((B, A)) < 0.7)
    mask[:, 2] = True
    legal = [np.nonzero(m)[0].tolist() for m in mask]
    noises = [rng.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal]
    res = []
    for no_lds in ("1", None):
        if no_lds:
            os.environ["LZ_TREE_NO_LDS"] = no_lds
        else:
            os.environ.pop("LZ_TREE_NO_LDS", None)
        roots = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S)
        roots.set_tiebreak(tiebreak, seed=99)
        model.initial_inference(obs, roots)
        roots.prepare_from_inference(0.25, noises, [-1] * B)
        L.check(L.lib().lz_roots_enable_trace(roots._h, 1))
        L.check(L.lib().lz_search(roots._h, S, 19652, 1.25, 0.997, 5, 0.01))
        tr = np.zeros((S, B, 4), np.int32)
        L.check(L.lib().lz_roots_read_trace(roots._h, S, tr.reshape(-1)))
        res.append((tr, roots.get_distributions(), np.asarray(roots.get_values(), np.float32).view(np.uint32),
                    roots.get_trajectories(), roots.get_minmax().view(np.uint32)))
    os.environ.pop("LZ_TREE_NO_LDS", None)
    assert np.array_equal(res[0][0], res[1][0]), "per-simulation records differ"
    assert res[0][1] == res[1][1] and np.array_equal(res[0][2], res[1][2]) and res[0][3] == res[1][3]
    assert np.array_equal(res[0][4], res[1][4])


def test_policy_with_device_select_action_and_reanalyze_shaped_batch():
    """(1) the collect / eval forward with select_action on the device returns the same eval actions
    and entropies as the Python original; (2) a reanalyze-shaped batch (batch_size x (unroll + 1) = 1536 roots, no
    exploration noise, game_buffer_efficientzero.py:325-409) goes through the same operator."""
    from oracle import torch_models as tm
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    from lightzero_amd.policy.efficientzero import EfficientZeroPolicy
    A = 6
    ref = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A))
    model = EfficientZeroModel(action_space_size=A).load_state_dict(ref.state_dict())
    cfg = dict(CFG, num_simulations=16, mcts_tiebreak="first")
    pol_host = EfficientZeroPolicy(cfg, model)
    pol_dev = EfficientZeroPolicy(dict(cfg, device_select_action=True), model)
    B = 32
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(21)).cuda()
    rng = np.random.default_rng(5)
    mask = (rng.random((B, A)) < 0.6).astype(np.float32); mask[:, 1] = 1
    e1 = pol_host._forward_eval(obs, action_mask=mask, to_play=[-1] * B)
    e2 = pol_dev._forward_eval(obs, action_mask=mask, to_play=[-1] * B)
    for i in range(B):
        assert e1[i]["action"] == e2[i]["action"] and e1[i]["visit_count_distributions"] == e2[i]["visit_count_distributions"]
        assert abs(e1[i]["visit_count_distribution_entropy"] - e2[i]["visit_count_distribution_entropy"]) < 1e-12
    c = pol_dev._forward_collect(obs, action_mask=mask, temperature=0.5, to_play=[-1] * B, epsilon=0.0)
    assert all(mask[i][c[i]["action"]] == 1 for i in range(B))
    # reanalyze-shaped batch
    R = 256 * 6
    obs_r = torch.rand(R, 4, 96, 96, generator=torch.Generator().manual_seed(22)).cuda()
    out = pol_host._forward_eval(obs_r, action_mask=np.ones((R, A), np.float32), to_play=[-1] * R)
    assert len(out) == R and all(sum(out[i]["visit_count_distributions"]) == 16 for i in range(R))
    # the first B rows of a bigger batch search exactly like a batch of their own (roots are independent)
    sub = pol_host._forward_eval(obs_r[:B].contiguous(), action_mask=np.ones((B, A), np.float32), to_play=[-1] * B)
    assert all(sub[i]["visit_count_distributions"] == out[i]["visit_count_distributions"] for i in range(B))


def test_search_with_reuse_fused_and_foreign_vs_oracle_pipeline():
    """ReZero: EfficientZeroMCTSCtree.search_with_reuse with (a) the engine model, whole loop on the
    device, and (b) a torch model driving the device tree through the reference loop, vs the oracle pipeline
    (C restatement of the reference tree + restated driver + torch model)."""
    from oracle import ctree as octree, search as osearch, torch_models as tm
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    from lightzero_amd.mcts.tree_search.mcts_ctree import EfficientZeroMCTSCtree
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    B, A, S = 32, 6, 24
    cfg = dict(CFG, num_simulations=S)
    ref = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A))
    model = EfficientZeroModel(action_space_size=A).load_state_dict(ref.state_dict())
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(31))
    rng = np.random.default_rng(7)
    legal = [list(range(A))] * B
    true_action = rng.integers(0, A, size=B).tolist()
    reuse_value = rng.standard_normal(B).astype(np.float32).tolist()
    with torch.no_grad():
        o = ref.initial_inference(obs)
    lat = o.latent_state.numpy(); rh = (o.reward_hidden_state[0].numpy(), o.reward_hidden_state[1].numpy())
    pol = o.policy_logits.numpy().tolist()
    # oracle pipeline
    oroots = octree.ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S)
    oroots.prepare_no_noise([0.0] * B, pol, [-1] * B)
    o_len, o_avg = osearch.ez_search_with_reuse(octree.ez_tree, oroots, ref, lat, rh, [-1] * B, cfg, true_action, reuse_value)
    assert o_avg < B  # some roots skipped inference
    mcts = EfficientZeroMCTSCtree(cfg)
    # (b) foreign torch model + device tree
    roots_b = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S)
    roots_b.set_tiebreak(0)
    roots_b.prepare_no_noise([0.0] * B, pol, [-1] * B)
    b_len, b_avg = mcts.search_with_reuse(roots_b, ref, lat, rh, [-1] * B, true_action, reuse_value)
    assert roots_b.get_distributions() == oroots.get_distributions() and (b_len, b_avg) == (o_len, o_avg)
    # (a) engine model, fused
    roots_a = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S)
    roots_a.set_tiebreak(0)
    out = model.initial_inference(obs.cuda().contiguous(), roots_a)
    roots_a.prepare_from_inference_no_noise([-1] * B)
    a_len, a_avg = mcts.search_with_reuse(roots_a, model, out.latent_state, out.reward_hidden_state, [-1] * B, true_action, reuse_value)
    same = sum(int(x == y) for x, y in zip(roots_a.get_distributions(), oroots.get_distributions()))
    assert same >= int(0.9 * B), "only %d / %d visit-count distributions identical" % (same, B)
    if same == B:
        assert (a_len, a_avg) == (o_len, o_avg)
    assert abs(a_avg - o_avg) < 0.1 * B


def test_packed_search_results_equal_the_individual_getters():
    from oracle import torch_models as tm
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    B, A, S = 16, 6, 12
    ref = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A))
    model = EfficientZeroModel(action_space_size=A).load_state_dict(ref.state_dict())
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(2)).cuda().contiguous()
    legal = [[0, 2, 5]] * 4 + [list(range(A))] * (B - 4)
    roots = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S)
    out = model.initial_inference(obs, roots)
    roots.prepare_from_inference_no_noise([-1] * B)
    L.check(L.lib().lz_search(roots._h, S, 19652, 1.25, 0.997, 5, 0.01))
    d = np.zeros((B, A), np.int32); c = np.zeros(B, np.int32); v = np.zeros(B, np.float32); p = np.zeros(B, np.float32)
    lg = np.zeros((B, A), np.float32)
    L.check(L.lib().lz_roots_get_search_results(roots._h, d, c, v, p.ctypes.data, lg.ctypes.data))
    assert [d[i, :c[i]].tolist() for i in range(B)] == roots.get_distributions()
    assert np.array_equal(v, np.asarray(roots.get_values(), np.float32))
    assert np.array_equal(p, out.value) and np.array_equal(lg, out.policy_logits)
    # the same read-back with select_action folded in == the separate calls (same temperature / mode / seed)
    for det, seed in ((True, 7), (False, 1234)):
        res = roots.get_search_results(select=(0.5, det, seed))
        pos, ent = roots.select_action(0.5, deterministic=det, seed=seed)
        assert np.array_equal(res[0], d) and np.array_equal(res[1], c) and np.array_equal(res[2], v)
        assert np.array_equal(res[5], pos) and np.array_equal(res[6], ent)


@pytest.mark.parametrize("tiebreak", [0, 1])
@pytest.mark.parametrize("family", ["ez", "mz"])
def test_tree_step_in_the_chain_prologue_is_bit_identical_to_the_separate_launch(family, tiebreak):
    """the graph-captured search runs a root's expand + backup + next selection as the prologue of the root's chain
    workgroup (k_chain<..., TREE>, tree code from lz_tree_dev.h compiled into a translation unit with FMA contraction
    on); LZ_NO_TREE_FUSE=1 keeps the separate k_backprop_traverse_lds launch.  Same distributions, root values
    (bitwise), trajectories and min-max statistics."""
    import os
    from oracle import torch_models as tm
    from lightzero_amd import _lib as L
    B, A, S = 67, 6, 50
    if family == "ez":
        from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree as tree
        from lightzero_amd.model.efficientzero_model import EfficientZeroModel as M
        ref = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A))
    else:
        from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree as tree
        from lightzero_amd.model.muzero_model import MuZeroModel as M
        ref = tm.synthetic_init(tm.MuZeroModel(action_space_size=A))
    model = M(action_space_size=A).load_state_dict(ref.state_dict())
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(12)).cuda().contiguous()
    rng = np.random.default_rng(4)
    mask = (rng.random((B, A)) < 0.7)
    mask[:, 1] = True
    legal = [np.nonzero(m)[0].tolist() for m in mask]
    noises = [rng.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal]
    res = []
    # (a) separate tree launch; (b) tree step in the chain prologue, head MLPs in their own launch (LZ_HEADS_LAUNCH=1): the same
    # arithmetic in another launch structure -> bit-identical; (c) the default: for EfficientZero the heads are SPLIT as well (first
    # layers in the LSTM launch, the rest in the next chain launch's prologue) -- their sums meet in another order, so the network
    # outputs move in the last bits and with them the root values; the exact replay gate (tests/test_exact_replay_gpu.py) holds
    # that path to the oracle on its own outputs
    for env in (dict(LZ_NO_TREE_FUSE="1"), dict(LZ_HEADS_LAUNCH="1"), {}):
        for k in ("LZ_NO_TREE_FUSE", "LZ_HEADS_LAUNCH"):
            os.environ.pop(k, None)
        os.environ.update(env)
        roots = tree.Roots(B, legal, action_space_size=A, max_simulations=S)
        roots.set_tiebreak(tiebreak, seed=77)
        model.initial_inference(obs, roots)
        roots.prepare_from_inference(0.25, noises, [-1] * B)
        L.check(L.lib().lz_search(roots._h, S, 19652, 1.25, 0.997, 5, 0.01))
        res.append((roots.get_distributions(), np.asarray(roots.get_values(), np.float32).view(np.uint32),
                    roots.get_trajectories(), roots.get_minmax().view(np.uint32)))
    for k in ("LZ_NO_TREE_FUSE", "LZ_HEADS_LAUNCH"):
        os.environ.pop(k, None)
    assert all(sum(d) == S for d in res[1][0]) and all(sum(d) == S for d in res[2][0])
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]
    assert np.array_equal(res[0][3], res[1][3])
    same = sum(int(a == b) for a, b in zip(res[1][0], res[2][0]))
    if family == "mz":
        assert same == B and np.array_equal(res[1][1], res[2][1])   # MuZero has no split heads: (c) is (b)
    else:
        assert same >= B - 2, "only %d / %d roots identical between the split-head and the head-launch path" % (same, B)
        v1, v2 = res[1][1].view(np.float32), res[2][1].view(np.float32)
        ok = np.array([a == b for a, b in zip(res[1][0], res[2][0])])
        assert np.abs(v1 - v2)[ok].max() < 2e-3


@pytest.mark.parametrize("tiebreak", [0, 1])
@pytest.mark.parametrize("family,A,ragged", [("ez", 6, True), ("ez", 6, False), ("mz", 4, False), ("ez", 8, True), ("mz", 5, True), ("ez", 3, True)])
def test_tree_parallel_selection_is_bit_identical_to_the_level_walk(family, A, ragged, tiebreak):
    """dev_traverse_par (every expanded node of a tree scored at once, lane = node; trees of <= 64 nodes, <= 8 actions) against
    dev_traverse (one level of the path at a time, lane = child; LZ_TRAVERSE_SERIAL=1) inside the same fused launch sequence:
    identical visit distributions, root values / min-max statistics (bitwise), trajectories and per-simulation records -- for
    the deterministic AND the stochastic tie-break (the draw is keyed by the node's depth in both), identity and ragged root legal
    lists, every compiled child count (4, 6, 8) with a smaller action space inside it, and sharp priors (deep paths)."""
    import os
    from oracle import torch_models as tm
    from lightzero_amd import _lib as L
    from lightzero_amd.model.synthetic import sharpen_state_dict
    B, S = 67, 50
    if family == "ez":
        from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree as tree
        from lightzero_amd.model.efficientzero_model import EfficientZeroModel as M
        ref = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A), seed=A)
    else:
        from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree as tree
        from lightzero_amd.model.muzero_model import MuZeroModel as M
        ref = tm.synthetic_init(tm.MuZeroModel(action_space_size=A), seed=A)
    sd = ref.state_dict()
    if A in (6, 5):
        sd = sharpen_state_dict(sd, 8.0)   # deep paths
    model = M(action_space_size=A).load_state_dict(sd)
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(40 + A)).cuda().contiguous()
    rng = np.random.default_rng(50 + A)
    mask = (rng.random((B, A)) < 0.7) if ragged else np.ones((B, A), bool)
    mask[:, A - 1] = True
    legal = [np.nonzero(m)[0].tolist() for m in mask]
    noises = [rng.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal]
    res = []
    for serial in (True, False):
        os.environ.pop("LZ_TRAVERSE_SERIAL", None)
        if serial:
            os.environ["LZ_TRAVERSE_SERIAL"] = "1"
        try:
            roots = tree.Roots(B, legal, action_space_size=A, max_simulations=S)
            roots.set_tiebreak(tiebreak, seed=91)
            model.initial_inference(obs, roots)   # (creates the device handle)
            L.check(L.lib().lz_roots_enable_trace(roots._h, 1))
            roots.prepare_from_inference(0.25, noises, [-1] * B)
            L.check(L.lib().lz_search(roots._h, S, 19652, 1.25, 0.997, 5 if family == "ez" else 0, 0.01))
            tr = np.zeros((S, B, 4), np.int32)
            L.check(L.lib().lz_roots_read_trace(roots._h, S, tr.reshape(-1)))
            res.append((roots.get_distributions(), np.asarray(roots.get_values(), np.float32).view(np.uint32),
                        roots.get_trajectories(), roots.get_minmax().view(np.uint32), tr))
        finally:
            os.environ.pop("LZ_TRAVERSE_SERIAL", None)
    a, b = res
    assert all(sum(d) == S for d in b[0])
    assert np.array_equal(a[4], b[4]), "per-simulation (slot, action, search length, to_play) records differ"
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2] == b[2] and np.array_equal(a[3], b[3])
    if A == 6:
        assert b[4][:, :, 2].max() >= 5   # the sharp prior did walk deep


@pytest.mark.parametrize("tiebreak", [0, 1])
@pytest.mark.parametrize("family,A,ragged,S,two_player", [("mz", 4, False, 300, False), ("ez", 6, True, 90, False), ("mz", 8, True, 120, True), ("ez", 3, True, 70, False)])
def test_workgroup_tree_step_is_bit_identical_to_the_one_wave_step(family, A, ragged, S, two_player, tiebreak):
    """k_tree_step_wg (deep trees: a workgroup per root, every expanded node scored at once, the walk follows stored choices / adds the
    mean-Q term for unvisited children) against k_backprop_traverse (one wave, level by level; LZ_TREE_NO_WG=1) -- the tree step kept
    out of the chain launch and out of LDS (LZ_NO_TREE_FUSE=1, LZ_TREE_NO_LDS=1) so that EVERY simulation runs the kernel under test:
    identical per-simulation records, visit distributions, root values and min-max statistics (bitwise), for both tie-break rules,
    ragged root lists, two players, trees of more than 256 nodes, sharp priors (deep paths)."""
    import os
    from oracle import torch_models as tm
    from lightzero_amd import _lib as L
    from lightzero_amd.model.synthetic import sharpen_state_dict
    B = 37
    if family == "ez":
        from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree as tree
        from lightzero_amd.model.efficientzero_model import EfficientZeroModel as M
        ref = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A), seed=20 + A)
    else:
        from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree as tree
        from lightzero_amd.model.muzero_model import MuZeroModel as M
        ref = tm.synthetic_init(tm.MuZeroModel(action_space_size=A), seed=20 + A)
    model = M(action_space_size=A).load_state_dict(sharpen_state_dict(ref.state_dict(), 6.0))
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(60 + A)).cuda().contiguous()
    rng = np.random.default_rng(70 + A)
    mask = (rng.random((B, A)) < 0.7) if ragged else np.ones((B, A), bool)
    mask[:, A - 1] = True
    legal = [np.nonzero(m)[0].tolist() for m in mask]
    noises = [rng.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal]
    to_play = rng.integers(1, 3, size=B).tolist() if two_player else [-1] * B
    res = []
    knobs = ("LZ_NO_TREE_FUSE", "LZ_TREE_NO_LDS", "LZ_TREE_NO_WG")
    for one_wave in (True, False):
        for k in knobs:
            os.environ.pop(k, None)
        os.environ["LZ_NO_TREE_FUSE"] = os.environ["LZ_TREE_NO_LDS"] = "1"
        if one_wave:
            os.environ["LZ_TREE_NO_WG"] = "1"
        try:
            roots = tree.Roots(B, legal, action_space_size=A, max_simulations=S)
            roots.set_tiebreak(tiebreak, seed=93)
            model.initial_inference(obs, roots)
            L.check(L.lib().lz_roots_enable_trace(roots._h, 1))
            roots.prepare_from_inference(0.25, noises, to_play)
            L.check(L.lib().lz_search(roots._h, S, 19652, 1.25, 0.997 if not two_player else 1.0, 5 if family == "ez" else 0, 0.01))
            tr = np.zeros((S, B, 4), np.int32)
            L.check(L.lib().lz_roots_read_trace(roots._h, S, tr.reshape(-1)))
            res.append((roots.get_distributions(), np.asarray(roots.get_values(), np.float32).view(np.uint32),
                        roots.get_trajectories(), roots.get_minmax().view(np.uint32), tr))
        finally:
            for k in knobs:
                os.environ.pop(k, None)
    a, b = res
    assert all(sum(d) == S for d in b[0])
    assert np.array_equal(a[4], b[4]), "per-simulation (slot, action, search length, to_play) records differ"
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2] == b[2] and np.array_equal(a[3], b[3])


def test_config2_full_size_deep_trees_properties():
    """BASELINE.json configs[2] at full size: Atari MuZero (conv), 1024 roots x 400 simulations, A = 4 -- the trees outgrow
    the LDS budget part-way through the search (tree step in the chain prologue -> separate HBM launch).  Size-independent
    properties: visit counts sum to S, every root value is finite, the search is idempotent (deterministic tie-break), and the
    run with the tree step kept out of the chain launch (LZ_NO_TREE_FUSE=1, LDS tree kernel -> HBM tree kernel) is
    bit-identical."""
    import os
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree
    from lightzero_amd.model.muzero_model import MuZeroModel
    from lightzero_amd.model.synthetic import muzero_state_dict
    B, A, S = 1024, 4, 400
    model = MuZeroModel(action_space_size=A).load_state_dict(muzero_state_dict(seed=0, action_space_size=A))
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(2)).cuda().contiguous()
    rng = np.random.default_rng(2)
    noises = rng.dirichlet([0.3] * A, size=B).astype(np.float32)
    legal = [list(range(A))] * B
    roots = mz_tree.Roots(B, legal, action_space_size=A, max_simulations=S)
    res = []
    for env in (None, None, "LZ_NO_TREE_FUSE", "LZ_TREE_NO_WG"):   # LZ_TREE_NO_WG: the one-wave HBM tree step instead of k_tree_step_wg
        if env:
            os.environ[env] = "1"
        roots.reset(legal)
        roots.set_tiebreak(0)
        model.initial_inference(obs, roots, fetch=False)
        roots.prepare_from_inference(0.25, noises, [-1] * B)
        L.check(L.lib().lz_search(roots._h, S, 19652, 1.25, 0.997, 0, 0.01))
        dist, cnt, val, pred, lg = roots.get_search_results()
        res.append((dist.copy(), val.copy().view(np.uint32)))
        for k in ("LZ_NO_TREE_FUSE", "LZ_TREE_NO_WG"):
            os.environ.pop(k, None)
    d0, v0 = res[0]
    assert (d0.sum(1) == S).all() and (d0 >= 0).all()
    assert np.isfinite(v0.view(np.float32)).all()
    for d, v in res[1:]:
        assert np.array_equal(d, d0) and np.array_equal(v, v0)


def test_reset_keep_inference_needs_a_fresh_root_inference():
    """lz_roots_reset_keep_inference re-arms roots whose representation network was ALREADY launched for this env-step; it is
    refused when no inference is pending (first use, or the last one was consumed by a prepare), and the reordered sequence
    inference -> reset(keep) -> prepare -> search gives the same result as reset -> inference -> prepare -> search."""
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    from lightzero_amd.model.synthetic import efficientzero_state_dict
    B, A, S = 24, 6, 16
    model = EfficientZeroModel(action_space_size=A).load_state_dict(efficientzero_state_dict(seed=0, action_space_size=A))
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(9)).cuda().contiguous()
    legal = [[0, 1, 4]] * 5 + [list(range(A))] * (B - 5)
    roots = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S)
    roots.set_tiebreak(0)
    with pytest.raises(L.LzError):
        roots.reset(legal, keep_inference=True)          # nothing inferred yet
    res = []
    for early in (False, True):
        if early:
            model.initial_inference(obs, roots, fetch=False)
            roots.reset(legal, keep_inference=True)
        else:
            roots.reset(legal)
            model.initial_inference(obs, roots, fetch=False)
        roots.prepare_from_inference_no_noise([-1] * B)
        L.check(L.lib().lz_search(roots._h, S, 19652, 1.25, 0.997, 5, 0.01))
        d, c, v, p, lg = roots.get_search_results()
        res.append((d.copy(), v.copy().view(np.uint32), p.copy().view(np.uint32)))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    with pytest.raises(L.LzError):
        roots.reset(legal, keep_inference=True)          # the inference was consumed by the prepare above
