"""Shared driver for the Sampled-EfficientZero tree parity tests (continuous actions): steps a module exposing the
surface of lzero/mcts/ctree/ctree_sampled_efficientzero/ezs_tree.pyx through S simulations with recorded (seeded)
network outputs, like SampledEfficientZeroMCTSCtree.search (lzero/mcts/tree_search/mcts_ctree_sampled.py)."""
import numpy as np

CASES = {
    "sez_cfg5_small": dict(B=8, D=1, K=20, S=50, seed=0),
    "sez_test_like": dict(B=5, D=2, K=6, S=100, seed=1, pb_c_base=1, pb_c_init=1.0, discount=0.9, delta=0.0,
                          noise_w=0.2),  # lzero/mcts/tests/test_mcts_sampled_ctree.py:61-137
    "sez_collide": dict(B=4, D=1, K=12, S=30, seed=2, sigma_scale=1e-7),  # tiny sigma: duplicate "%f" keys
    "sez_2p": dict(B=6, D=3, K=8, S=40, seed=3, to_play="random12", discount=1.0),
    "sez_cfg5_b256": dict(B=256, D=1, K=20, S=50, seed=4),
    # discrete action spaces (continuous_action_space=False): K of the A actions, policy = A logits, D = 1 (the action index)
    "sez_disc_a6_k4": dict(B=8, D=1, K=4, S=40, seed=5, A=6),
    "sez_disc_pendulum_a11_k5": dict(B=6, D=1, K=5, S=50, seed=6, A=11),  # pendulum_cont_disc_sampled_efficientzero_config.py
    "sez_disc_k_eq_a": dict(B=4, D=1, K=4, S=30, seed=7, A=4),
}


def make_inputs(case):
    c = dict(pb_c_base=19652, pb_c_init=1.25, discount=0.997, delta=0.01, noise_w=0.25, horizon=5, to_play=None,
             sigma_scale=1.0)
    c.update(case)
    rng = np.random.default_rng(100 + c["seed"])
    B, D, K, S = c["B"], c["D"], c["K"], c["S"]

    def policy():
        if c.get("A"):
            return rng.standard_normal((B, c["A"])).astype(np.float32)
        mu = 0.5 * rng.standard_normal((B, D))
        sigma = (0.3 + rng.random((B, D))) * c["sigma_scale"]
        return np.concatenate([mu, sigma], 1).astype(np.float32)
    to_play = [-1] * B if c["to_play"] is None else rng.integers(1, 3, size=B).tolist()
    noises = None if c["noise_w"] is None else rng.dirichlet([0.3] * K, size=B).astype(np.float32)
    sims = [dict(vp=(0.5 * rng.standard_normal(B)).astype(np.float32), v=rng.standard_normal(B).astype(np.float32),
                 policy=policy()) for _ in range(S)]
    c.update(to_play_list=to_play, noises=noises, root_policy=policy(), root_vp=np.zeros(B, np.float32), sims=sims)
    return c


def run_tree(mod, c, make_roots, before_expand=None, after_expand=None):
    """make_roots() -> Roots;  before_expand(roots, record_index) is called before every expand (prepare = record 0,
    simulation s = record s + 1) so that a harness can set the clock or inject samples; after_expand likewise."""
    B, S = c["B"], c["S"]
    roots = make_roots()
    if before_expand:
        before_expand(roots, 0)
    if c["noises"] is not None:
        roots.prepare(c["noise_w"], c["noises"].tolist(), c["root_vp"].tolist(), c["root_policy"].tolist(), list(c["to_play_list"]))
    else:
        roots.prepare_no_noise(c["root_vp"].tolist(), c["root_policy"].tolist(), list(c["to_play_list"]))
    if after_expand:
        after_expand(roots, 0)
    mm = mod.MinMaxStatsList(B)
    mm.set_delta(c["delta"])
    rec = np.zeros((S, B, 4), np.int32)
    last = np.zeros((S, B, c["D"]), np.float32)
    for s in range(S):
        res = mod.ResultsWrapper(B)
        ix, iy, la, vtp = mod.batch_traverse(roots, c["pb_c_base"], c["pb_c_init"], c["discount"], mm, res,
                                             list(c["to_play_list"]), not c.get("A"))
        sl = res.get_search_len()
        rec[s, :, 0], rec[s, :, 1], rec[s, :, 2], rec[s, :, 3] = ix, iy, sl, vtp
        last[s] = np.asarray(la, np.float32).reshape(B, c["D"])
        sim = c["sims"][s]
        if before_expand:
            before_expand(roots, s + 1)
        mod.batch_backpropagate(s + 1, c["discount"], sim["vp"].tolist(), sim["v"].tolist(), sim["policy"].tolist(), mm,
                                res, [int(l % c["horizon"] == 0) for l in sl], vtp)
        if after_expand:
            after_expand(roots, s + 1)
    return dict(records=rec, last_actions=last, distributions=np.asarray(roots.get_distributions(), np.int32),
                values=np.asarray(roots.get_values(), np.float32),
                root_actions=np.asarray(roots.get_sampled_actions(), np.float32).reshape(B, c["K"], c["D"]))


def assert_same(a, b, what=""):
    assert np.array_equal(a["root_actions"].view(np.uint32), b["root_actions"].view(np.uint32)), "%s: root sampled actions differ" % what
    assert np.array_equal(a["records"], b["records"]), "%s: per-simulation (ix, iy, len, to_play) differ" % what
    assert np.array_equal(a["last_actions"].view(np.uint32), b["last_actions"].view(np.uint32)), "%s: last actions differ" % what
    assert np.array_equal(a["distributions"], b["distributions"]), "%s: visit-count distributions differ" % what
    assert np.array_equal(a["values"].view(np.uint32), b["values"].view(np.uint32)), "%s: root values differ" % what
