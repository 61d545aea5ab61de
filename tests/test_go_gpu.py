"""GPU: BASELINE config 4 shape -- Go 9x9 MuZero: obs 17x9x9 (zoo/board_games/go/envs/go_env.py:49), no downsample,
latent 64x9x9, action space 82 (81 points + pass), two players, random legal masks -- fused search vs the oracle
pipeline, plus the full-size (64 roots per GPU x 200 simulations) run checked through size-independent properties."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = dict(num_simulations=30, pb_c_base=19652, pb_c_init=1.25, discount_factor=1.0, value_delta_max=0.01,
           root_noise_weight=0.25, root_dirichlet_alpha=0.3)
A = 82
# (observation_shape, action space): Go 9x9, and Connect4 (zoo/board_games/connect4/config/connect4_muzero_bot_mode_config.py:
# 31-35: obs 3x6x7, 7 columns, 64 channels) -- a non-square grid with a padded last M-tile (42 = 2 x 16 + 10 pixels)
GAMES = {"go": ((17, 9, 9), 82), "connect4": ((3, 6, 7), 7)}
# measured (profiles/r06_parity.json, e2e/mz_go/*, e2e/mz_connect4/*); the gate sits one root below the measurement
GATE_E2E = {"go": 0.95, "connect4": 0.95}


def _setup(B, seed=0, game="go"):
    from oracle import torch_models as tm
    from lightzero_amd.model.muzero_model import MuZeroModel
    shape, A = GAMES[game]
    kw = dict(observation_shape=shape, action_space_size=A, downsample=False)
    ref = tm.synthetic_init(tm.MuZeroModel(**kw), seed=seed)
    dev = MuZeroModel(**kw).load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(seed + 1)
    obs = (torch.rand(B, *shape, generator=g) < 0.3).float()
    rng = np.random.default_rng(seed)
    legal = []
    for _ in range(B):
        m = rng.random(A) < 0.7
        m[A - 1] = True
        legal.append(np.nonzero(m)[0].tolist())
    to_play = rng.integers(1, 3, size=B).tolist()
    noises = [rng.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal]
    return ref, dev, obs, legal, to_play, noises


@pytest.mark.parametrize("game", sorted(GAMES))
def test_go_fused_search_vs_oracle(game):
    from oracle import ctree as octree, search as osearch, torch_models as tm
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree
    B, S = 64, CFG["num_simulations"]
    (_, GH, GW), A = GAMES[game]
    ref, model, obs, legal, to_play, noises = _setup(B, game=game)
    lib = L.lib()
    roots = mz_tree.Roots(B, legal, action_space_size=A, max_simulations=S)
    roots.set_tiebreak(0)
    out = model.initial_inference(obs.cuda().contiguous(), roots)
    with torch.no_grad():
        o = ref.initial_inference(obs)
    lat0 = np.zeros((B, 64, GH, GW), np.float32)
    L.check(lib.lz_roots_read_latent(roots._h, 0, lat0.reshape(-1)))
    import parity_record
    def rel0(a, b):
        return float((np.abs(np.asarray(a, np.float64) - b) / (1.0 + np.abs(b))).max())
    parity_record.check("initial_inference/mz_%s/B%d" % (game, B), dict(latent=rel0(lat0, o.latent_state.numpy().astype(np.float64)),
                                                                     policy=rel0(out.policy_logits, o.policy_logits.numpy().astype(np.float64))))
    roots.prepare_from_inference(CFG["root_noise_weight"], noises, to_play)
    L.check(lib.lz_roots_enable_trace(roots._h, 1))
    L.check(lib.lz_search(roots._h, S, CFG["pb_c_base"], CFG["pb_c_init"], CFG["discount_factor"], 0, CFG["value_delta_max"]))
    trace = np.zeros((S, B, 4), np.int32)
    L.check(lib.lz_roots_read_trace(roots._h, S, trace.reshape(-1)))
    lat = np.zeros((S + 1, B, 64, GH, GW), np.float32)
    rew = np.zeros((S + 1, B), np.float32); val = np.zeros_like(rew); pol = np.zeros((S + 1, B, A), np.float32)
    for s in range(S + 1):
        L.check(lib.lz_roots_read_latent(roots._h, s, lat[s].reshape(-1)))
        L.check(lib.lz_roots_read_sim_outputs(roots._h, s, rew[s], val[s], pol[s].reshape(-1)))
    ist = tm.InverseScalarTransform()
    ar = np.arange(B)

    def rel(a, b):
        return float((np.abs(a - b) / (1.0 + np.abs(b))).max())
    # every simulation also in binary64 on the same teacher-forced inputs: no reset bounds these latents (they grow with the depth of the
    # path), so torch's own fp32 result is several 1e-6 from the exact value of the network here -- parity_record.check_vs_truth
    import copy
    ref64 = copy.deepcopy(ref).double()
    worst = dict(lat=0.0, pol=0.0, rew=0.0, val=0.0)
    d64, t64 = dict(latent=0.0, policy=0.0), dict(latent=0.0, policy=0.0)
    for s in range(S):
        ix, act = trace[s, :, 0], trace[s, :, 1]
        with torch.no_grad():
            q = ref.recurrent_inference(torch.from_numpy(lat[ix, ar]), torch.from_numpy(act).long())
            q64 = ref64.recurrent_inference(torch.from_numpy(lat[ix, ar]).double(), torch.from_numpy(act).long())
        worst["lat"] = max(worst["lat"], rel(lat[s + 1], q.latent_state.numpy()))
        worst["pol"] = max(worst["pol"], rel(pol[s + 1], q.policy_logits.numpy()))
        worst["rew"] = max(worst["rew"], rel(rew[s + 1], ist(q.reward).reshape(-1).numpy()))
        worst["val"] = max(worst["val"], rel(val[s + 1], ist(q.value).reshape(-1).numpy()))
        l64, p64 = q64.latent_state.numpy(), q64.policy_logits.numpy()
        d64["latent"] = max(d64["latent"], rel(lat[s + 1].astype(np.float64), l64)); d64["policy"] = max(d64["policy"], rel(pol[s + 1].astype(np.float64), p64))
        t64["latent"] = max(t64["latent"], rel(q.latent_state.numpy().astype(np.float64), l64)); t64["policy"] = max(t64["policy"], rel(q.policy_logits.numpy().astype(np.float64), p64))
    print("go worst |d| / (1 + |ref|):", worst, "device vs binary64:", d64, "torch fp32 vs binary64:", t64)
    import parity_record
    name = "recurrent_teacher_forced/mz_%s/B%d_S%d" % (game, B, S)
    parity_record.check_vs_truth(name, dict(latent=worst["lat"], policy=worst["pol"]), d64, t64, extra=dict(batch=B, simulations=S))
    parity_record.check(name, dict(reward=worst["rew"], value=worst["val"]))
    rec_o = []
    o_dist, o_val, _, o_logits = osearch.mz_forward_collect(octree.mz_tree, ref, obs, legal, noises, to_play, CFG,
                                                            roots_kwargs=dict(action_space_size=A, max_simulations=S),
                                                            deterministic=True, record=rec_o)
    d_dist, d_val = roots.get_distributions(), np.array(roots.get_values())
    assert [len(d) for d in d_dist] == [len(l) for l in legal]
    # two-player searches: recorded + every differing root attributed (tests/e2e_common.py); gated at the evidence, not at 0.8
    import e2e_common
    e2e_common.attribute_and_gate("e2e/mz_%s/B%d_S%d" % (game, B, S), "mz", octree.mz_tree, CFG, A, legal, noises, to_play, o_logits,
                                  np.asarray(out.policy_logits, np.float32), rec_o, e2e_common.device_records(roots, lib, L, B, A, S),
                                  o_dist, d_dist, o_val, d_val, gate=GATE_E2E[game])


def test_go_full_size_properties():
    """64 roots (512 envs over 8 GPUs) x 200 simulations: visit counts sum to S, only legal actions are visited,
    root values finite, deterministic tie-break => two runs are identical (idempotence)."""
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree
    B, S = 64, 200
    ref, model, obs, legal, to_play, noises = _setup(B, seed=3)
    d_obs = obs.cuda().contiguous()
    res = []
    for _ in range(2):
        roots = mz_tree.Roots(B, legal, action_space_size=A, max_simulations=S)
        roots.set_tiebreak(0)
        model.initial_inference(d_obs, roots)
        roots.prepare_from_inference(0.25, noises, to_play)
        L.check(L.lib().lz_search(roots._h, S, 19652, 1.25, 1.0, 0, 0.01))
        res.append((roots.get_distributions(), roots.get_values(), roots.get_trajectories()))
    (d1, v1, t1), (d2, v2, t2) = res
    assert d1 == d2 and v1 == v2 and t1 == t2
    assert all(sum(d) == S for d in d1) and [len(d) for d in d1] == [len(l) for l in legal]
    assert np.isfinite(np.array(v1)).all()
    assert all(t[0] in l for t, l in zip(t1, legal))
