"""Shared driver for the Gumbel MuZero tree parity tests: steps any module exposing the surface of
lzero/mcts/ctree/ctree_gumbel_muzero/gmz_tree.pyx through S simulations with recorded (seeded random) network outputs the
way GumbelMuZeroMCTSCtree.search does (lzero/mcts/tree_search/mcts_ctree.py:1067-1172) and records what is observable."""
import numpy as np

from tree_driver import fixture_legal_actions

CASES = {
    "gmz_b16_a6": dict(B=16, A=6, S=50, m=4, seed=40),
    "gmz_cfg_like_b64_a18": dict(B=64, A=18, S=50, m=16, seed=41),
    "gmz_fixture16": dict(B=16, A=9, S=30, m=4, seed=42, legal="fixture"),
    "gmz_few_sims": dict(B=8, A=4, S=5, m=8, seed=43, noise_w=None),
    "gmz_zero_ties": dict(B=4, A=5, S=20, m=4, seed=44, zero=True, noise_w=None),
    "gmz_m1": dict(B=4, A=6, S=12, m=1, seed=45),
    # action spaces beyond 256: the kernels' 8- and 16-chunk instances (512 / 1024 actions)
    "gmz_wide_a300": dict(B=6, A=300, S=40, m=16, seed=46, legal="random"),
    "gmz_wide_a1000": dict(B=4, A=1000, S=50, m=8, seed=47),
}


def make_inputs(case):
    c = dict(discount=0.997, delta=0.01, noise_w=0.25, legal=None, zero=False)
    c.update(case)
    rng = np.random.default_rng(c["seed"])
    B, A, S = c["B"], c["A"], c["S"]
    if c["legal"] == "fixture":
        legal = fixture_legal_actions()
        assert len(legal) == B, "the fixture's legal lists are for %d roots" % len(legal)
    elif c["legal"] == "random":      # ragged lists, at least one action each (a generator of its own: the other arrays do not move)
        lr = np.random.default_rng(c["seed"] + 77777)
        legal = []
        for _ in range(B):
            k = lr.random(A) < 0.5
            k[lr.integers(0, A)] = True
            legal.append(np.nonzero(k)[0].tolist())
    else:
        legal = [list(range(A)) for _ in range(B)]
    z = 0.0 if c["zero"] else 1.0
    c["legal_list"] = legal
    c["root_logits"] = (z * rng.standard_normal((B, A))).astype(np.float32)
    c["root_reward"] = (z * 0.1 * rng.standard_normal(B)).astype(np.float32)
    c["root_value"] = (z * rng.standard_normal(B)).astype(np.float32)
    c["noises"] = None if c["noise_w"] is None else [rng.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal]
    c["sims"] = [dict(r=(z * 0.5 * rng.standard_normal(B)).astype(np.float32), v=(z * rng.standard_normal(B)).astype(np.float32),
                      logits=(z * rng.standard_normal((B, A))).astype(np.float32)) for _ in range(S)]
    return c


def run_tree(mod, c, roots_kwargs=None):
    B, S, A = c["B"], c["S"], c["A"]
    roots = mod.Roots(B, c["legal_list"], **(roots_kwargs or {}))
    tp = [-1] * B
    if c["noises"] is not None:
        roots.prepare(c["noise_w"], c["noises"], c["root_reward"].tolist(), c["root_value"].tolist(), c["root_logits"].tolist(), tp)
    else:
        roots.prepare_no_noise(c["root_reward"].tolist(), c["root_value"].tolist(), c["root_logits"].tolist(), tp)
    mm = mod.MinMaxStatsList(B)
    mm.set_delta(c["delta"])
    rec = np.zeros((S, B, 4), np.int32)
    for s in range(S):
        res = mod.ResultsWrapper(B)
        ix, iy, la, vtp = mod.batch_traverse(roots, S, c["m"], c["discount"], res, list(tp))
        rec[s, :, 0], rec[s, :, 1], rec[s, :, 2], rec[s, :, 3] = ix, iy, la, res.get_search_len()
        sim = c["sims"][s]
        mod.batch_back_propagate(s + 1, c["discount"], sim["r"].tolist(), sim["v"].tolist(), sim["logits"].tolist(), mm, res, vtp)
    dist = np.full((B, A), -1, np.int32)
    for i, d in enumerate(roots.get_distributions()):
        dist[i, :len(d)] = d
    return dict(records=rec, distributions=dist, values=np.asarray(roots.get_values(), np.float32),
                policies=np.asarray(roots.get_policies(c["discount"], A), np.float32),
                children_values=np.asarray(roots.get_children_values(c["discount"], A), np.float32))


def assert_same(a, b, what=""):
    assert np.array_equal(a["records"], b["records"]), "%s: per-simulation (ix, iy, action, len) differ" % what
    assert np.array_equal(a["distributions"], b["distributions"]), "%s: visit counts differ" % what
    for k in ("values", "policies", "children_values"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), "%s: %s differ" % (what, k)
