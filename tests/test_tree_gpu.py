"""GPU parity (through the C ABI): the HIP tree kernels vs the CPU oracle and the golden vectors
generated from the compiled reference.  Bit-exact: integer records and visit counts identical,
root values identical to the last bit (deterministic first-arg-max tie-break on both sides)."""
import os

import numpy as np
import pytest

import tree_driver as td

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _dev_mod(variant):
    if variant == "ez":
        from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree as m
    else:
        from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree as m
    return m


def _run_dev(c):
    class Det(object):  # Roots with the deterministic tie-break selected before prepare
        pass
    mod = _dev_mod(c["variant"])
    orig = mod.Roots

    def mk(n, legal, **kw):
        r = orig(n, legal, action_space_size=c["A"], max_simulations=c["S"])
        r.set_tiebreak(0)
        return r
    ns = type("M", (), dict(Roots=staticmethod(mk), MinMaxStatsList=mod.MinMaxStatsList,
                            ResultsWrapper=mod.ResultsWrapper, batch_traverse=staticmethod(mod.batch_traverse),
                            batch_backpropagate=staticmethod(mod.batch_backpropagate)))
    return td.run_tree(ns, c)


@pytest.mark.parametrize("name", sorted(td.CASES))
def test_device_tree_matches_oracle_and_golden(name):
    from oracle import ctree as octree
    c = td.make_inputs(td.CASES[name])
    dev = _run_dev(c)
    omod = octree.ez_tree if c["variant"] == "ez" else octree.mz_tree
    ora = td.run_tree(omod, c, roots_kwargs=dict(action_space_size=c["A"], max_simulations=c["S"]))
    td.assert_same(ora, dev, name)
    assert np.array_equal(ora["minmax"].view(np.uint32), dev["minmax"].view(np.uint32)), "min/max stats differ"
    g = np.load(os.path.join(GOLD, "tree_%s.npz" % name))
    assert np.array_equal(dev["records"], g["records"])
    assert np.array_equal(dev["values"].view(np.uint32), g["values"].view(np.uint32))


def test_reference_known_answer_deterministic_first_action_device():
    # lzero/mcts/tests/test_muzero_ctree_deterministic.py:4-25
    from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree
    roots = mz_tree.Roots(1, [[0, 1, 2]])
    roots.prepare_no_noise([0.0], [[0.0, 0.0, 0.0]], [-1])
    mm = mz_tree.MinMaxStatsList(1)
    mm.set_delta(0.01)
    sel = []
    for _ in range(5):
        res = mz_tree.ResultsWrapper(1)
        sel.append(mz_tree.batch_traverse(roots, 19652, 1.25, 0.997, mm, res, [-1], deterministic=True)[2][0])
    assert sel == [0] * 5


def test_reference_stochastic_tie_breaking_device():
    # lzero/mcts/tests/test_muzero_ctree_deterministic.py:28-48
    from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree
    roots = mz_tree.Roots(1, [[0, 1, 2]])
    roots.prepare_no_noise([0.0], [[0.0, 0.0, 0.0]], [-1])
    mm = mz_tree.MinMaxStatsList(1)
    mm.set_delta(0.01)
    sel = []
    for _ in range(30):
        res = mz_tree.ResultsWrapper(1)
        sel.append(mz_tree.batch_traverse(roots, 19652, 1.25, 0.997, mm, res, [-1])[2][0])
    assert len(set(sel)) > 1


def test_random_tiebreak_is_uniform_over_tie_list():
    # all-zero network => every score ties; stochastic mode must spread root visits uniformly-ish
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    B, A, S = 64, 6, 60
    roots = ez_tree.Roots(B, [list(range(A))] * B, action_space_size=A, max_simulations=S)
    roots.prepare_no_noise([0.0] * B, np.zeros((B, A), np.float32).tolist(), [-1] * B)
    mm = ez_tree.MinMaxStatsList(B)
    mm.set_delta(0.01)
    first = np.zeros(A, np.int64)
    for s in range(S):
        res = ez_tree.ResultsWrapper(B)
        ix, iy, la, vtp = ez_tree.batch_traverse(roots, 19652, 1.25, 0.997, mm, res, [-1] * B)
        if s == 0:
            for a in la:
                first[a] += 1
        sl = res.get_search_len()
        ez_tree.batch_backpropagate(s + 1, 0.997, [0.0] * B, [0.0] * B, np.zeros((B, A), np.float32).tolist(), mm, res,
                                    [int(l % 5 == 0) for l in sl], vtp)
    assert (first > 0).all(), first  # 64 draws over 6 tied actions: every action picked at least once
    dist = np.array(roots.get_distributions())
    assert (dist.sum(1) == S).all()


def test_shapes_and_legality_like_reference_tests():
    # lzero/mcts/tests/test_mcts_ctree.py:180-181,272-283: distribution lengths == #legal actions
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    legal = td.fixture_legal_actions()
    c = td.make_inputs(td.CASES["ez_fixture16"])
    out = td.run_tree(ez_tree, c)
    assert [len(d) for d in out["distributions"]] == [len(l) for l in legal]
    assert all(sum(d) == c["S"] for d in out["distributions"])


def test_error_paths_are_loud():
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    roots = ez_tree.Roots(2, [[0, 1], [0, 1]], action_space_size=2, max_simulations=3)
    roots.prepare_no_noise([0.0, 0.0], [[0.0, 0.0], [0.0, 0.0]], [-1, -1])
    mm = ez_tree.MinMaxStatsList(2)
    mm.set_delta(0.01)
    res = ez_tree.ResultsWrapper(2)
    _, _, _, vtp = ez_tree.batch_traverse(roots, 19652, 1.25, 0.997, mm, res, [-1, -1])
    with pytest.raises(L.LzError):  # node pool sized for 3 simulations
        ez_tree.batch_backpropagate(4, 0.997, [0.0, 0.0], [0.0, 0.0], [[0.0, 0.0]] * 2, mm, res, [0, 0], vtp)
    with pytest.raises(L.LzError):
        bad = ez_tree.Roots(1, [[5]], action_space_size=2, max_simulations=3)
        bad.prepare_no_noise([0.0], [[0.0, 0.0]], [-1])


def test_device_select_action_matches_python_original():
    """lz_roots_select_action vs lzero/policy/utils.py:637-661 (restated in lightzero_amd.policy.utils): arg-max position
    identical, entropy (float64) within 1e-12, sampled positions follow N^(1/T)."""
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    from lightzero_amd.policy.utils import select_action
    rng = np.random.default_rng(0)
    B, A, S = 64, 9, 40
    legal = []
    for i in range(B):
        k = int(rng.integers(1, A + 1))
        legal.append(sorted(rng.choice(A, size=k, replace=False).tolist()))
    roots = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S)
    roots.set_tiebreak(0)
    noises = [rng.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal]
    roots.prepare(0.25, noises, [0.0] * B, rng.normal(size=(B, A)).astype(np.float32).tolist(), [-1] * B)
    mm = ez_tree.MinMaxStatsList(B); mm.set_delta(0.01)
    for s in range(S):
        res = ez_tree.ResultsWrapper(B)
        ix, iy, la, vtp = ez_tree.batch_traverse(roots, 19652, 1.25, 0.997, mm, res, [-1] * B)
        sl = res.get_search_len()
        ez_tree.batch_backpropagate(s + 1, 0.997, rng.normal(size=B).astype(np.float32).tolist(), rng.normal(size=B).astype(np.float32).tolist(),
                                    rng.normal(size=(B, A)).astype(np.float32).tolist(), mm, res, [int(l % 5 == 0) for l in sl], vtp)
    dists = roots.get_distributions()
    for T in (1.0, 0.5, 0.25):
        pos, ent = roots.select_action(T, deterministic=True)
        for i in range(B):
            p_ref, e_ref = select_action(dists[i], temperature=T, deterministic=True)
            assert pos[i] == p_ref
            assert abs(ent[i] - e_ref) <= 1e-12 * max(1.0, abs(e_ref))
    # sampling: empirical frequencies of root 0 over many seeds vs N^(1/T)
    T = 1.0
    i = int(np.argmax([len(d) for d in dists]))
    p = np.asarray(dists[i], np.float64) ** (1 / T); p /= p.sum()
    draws = np.stack([roots.select_action(T, deterministic=False, seed=1000 + k)[0] for k in range(400)])
    freq = np.bincount(draws[:, i], minlength=len(p)) / draws.shape[0]
    assert np.abs(freq - p).max() < 0.08
    for b in range(B):
        assert draws[:, b].max() < len(dists[b]) and all(dists[b][j] > 0 for j in np.unique(draws[:, b]))


@pytest.mark.parametrize("name", sorted(td.REUSE_CASES))
def test_device_reuse_tree_matches_oracle_and_golden(name):
    """ReZero batch_traverse_with_reuse / batch_backpropagate_with_reuse on the device: bit-exact vs the C oracle and vs
    the goldens generated from the reference's own compiled code (incl. the packed batch_index bookkeeping)."""
    from oracle import ctree as octree
    c = td.make_reuse_inputs(td.REUSE_CASES[name])
    mod = _dev_mod(c["variant"])
    orig = mod.Roots

    def mk(n, legal, **kw):
        r = orig(n, legal, action_space_size=c["A"], max_simulations=c["S"])
        r.set_tiebreak(0)
        return r
    ns = type("M", (), dict(Roots=staticmethod(mk), MinMaxStatsList=mod.MinMaxStatsList, ResultsWrapper=mod.ResultsWrapper,
                            batch_traverse_with_reuse=staticmethod(mod.batch_traverse_with_reuse),
                            batch_backpropagate_with_reuse=staticmethod(mod.batch_backpropagate_with_reuse)))
    dev = td.run_tree_reuse(ns, c)
    omod = octree.ez_tree if c["variant"] == "ez" else octree.mz_tree
    ora = td.run_tree_reuse(omod, c, roots_kwargs=dict(action_space_size=c["A"], max_simulations=c["S"]))
    td.assert_same(ora, dev, name)
    assert dev["inferences"] == ora["inferences"]
    g = np.load(os.path.join(GOLD, "tree_%s.npz" % name))
    assert np.array_equal(dev["records"], g["records"])
    assert np.array_equal(dev["values"].view(np.uint32), g["values"].view(np.uint32))
