"""GPU: randomised sweep of the fine-grained tree surface (Roots.prepare / batch_traverse / batch_backpropagate through the C ABI)
against the oracle trees -- 48 seeded configurations (+ 16 of the ReZero reuse surface) over both tree variants, 1..48 roots, 1..70 actions (full and ragged legal
lists), 1..64 simulations, one and two players, discounts, pb_c constants, min-max deltas, noise weights, LSTM horizons and logit
scales.  Bit-exact like tests/test_tree_gpu.py: selection records, visit counts, root values, min-max statistics."""
import numpy as np
import pytest

import tree_driver as td
from test_tree_gpu import _dev_mod, _run_dev

# LZ_FUZZ_SEED_OFFSET=n shifts every seeded sweep of this file to seeds n .. n + count - 1 (ad-hoc wider sweeps; the committed suite runs 0)
_OFF = int(__import__("os").environ.get("LZ_FUZZ_SEED_OFFSET", "0"))

pytestmark = pytest.mark.gpu


def _case(seed):
    r = np.random.default_rng(9000 + seed)
    two = bool(r.integers(0, 2))
    return dict(variant=["ez", "mz"][int(r.integers(0, 2))], B=int(r.integers(1, 49)), A=int(r.integers(1, 71)), S=int(r.integers(1, 65)),
                seed=100 + seed, legal=[None, "random"][int(r.integers(0, 2))], to_play="random12" if two else None,
                discount=float(r.choice([0.997, 1.0, 0.9])), pb_c_base=int(r.choice([19652, 1, 100])), pb_c_init=float(r.choice([1.25, 0.5, 2.0])),
                delta=float(r.choice([0.01, 0.0, 0.1])), noise_w=[0.25, None, 0.5][int(r.integers(0, 3))], horizon=int(r.choice([5, 1, 3])),
                scale=float(r.choice([1.0, 5.0, 0.1])), zero=bool(r.random() < 0.1))


@pytest.mark.parametrize("seed", range(_OFF, _OFF + 48))
def test_random_configuration_matches_the_oracle(seed):
    from oracle import ctree as octree
    case = _case(seed)
    c = td.make_inputs(case)
    dev = _run_dev(c)
    omod = octree.ez_tree if c["variant"] == "ez" else octree.mz_tree
    ora = td.run_tree(omod, c, roots_kwargs=dict(action_space_size=c["A"], max_simulations=c["S"]))
    td.assert_same(ora, dev, repr(case))
    assert np.array_equal(ora["minmax"].view(np.uint32), dev["minmax"].view(np.uint32)), "min/max stats differ: %r" % (case,)


@pytest.mark.parametrize("seed", range(_OFF, _OFF + 16))
def test_random_reuse_configuration_matches_the_oracle(seed):
    """ReZero: batch_traverse_with_reuse / batch_backpropagate_with_reuse (some roots skip inference) on random configurations"""
    from oracle import ctree as octree
    case = _case(500 + seed)
    case["A"] = max(case["A"], 2)
    c = td.make_reuse_inputs(case)
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
    td.assert_same(ora, dev, repr(case))
    assert dev["inferences"] == ora["inferences"]


@pytest.mark.parametrize("seed", range(_OFF, _OFF + 16))
def test_random_sampled_tree_configuration_matches_the_oracle(seed):
    """Sampled EfficientZero tree (continuous and discrete action spaces) with the oracle's draws injected"""
    import sampled_driver as sd
    from oracle import ctree as octree
    from lightzero_amd.mcts.ctree.ctree_sampled_efficientzero import ezs_tree
    r = np.random.default_rng(7000 + seed)
    disc = bool(r.integers(0, 2))
    K = int(r.integers(2, 24))
    case = dict(B=int(r.integers(1, 24)), D=1 if disc else int(r.integers(1, 4)), K=K, S=int(r.integers(1, 48)), seed=300 + seed,
                discount=float(r.choice([0.997, 1.0, 0.9])), delta=float(r.choice([0.01, 0.0])), noise_w=float(r.choice([0.25, 0.5])),
                pb_c_base=int(r.choice([19652, 1])), pb_c_init=float(r.choice([1.25, 1.0])))
    if disc:
        case["A"] = K + int(r.integers(0, 6))
    if r.random() < 0.4:
        case["to_play"] = "random12"
    c = sd.make_inputs(case)
    draws = {}

    def mk_o():
        o = octree.ezs_tree.Roots(c["B"], [[-1] * 5] * c["B"], c.get("A") or c["D"], c["K"], not c.get("A"), max_simulations=c["S"])
        o.set_clock(123456789)
        return o
    ora = sd.run_tree(octree.ezs_tree, c, mk_o, after_expand=lambda o, e: draws.__setitem__(e, np.asarray(o.get_sampled_actions(e), np.float32)))

    def mk_d():
        d = ezs_tree.Roots(c["B"], [[-1] * 5] * c["B"], c.get("A") or c["D"], c["K"], not c.get("A"), max_simulations=c["S"])
        d.set_tiebreak(0)
        return d
    dev = sd.run_tree(ezs_tree, c, mk_d, before_expand=lambda d, e: setattr(d, "given", draws[e]))
    sd.assert_same(ora, dev, repr(case))


@pytest.mark.parametrize("seed", range(_OFF, _OFF + 16))
def test_random_gumbel_tree_configuration_matches_the_oracle(seed):
    import gumbel_driver as gd
    from oracle import ctree as octree
    from test_gumbel_gpu import _dev_mod as gdev
    r = np.random.default_rng(8000 + seed)
    A = int(r.integers(2, 40))
    case = dict(B=int(r.integers(1, 40)), A=A, S=int(r.integers(1, 64)), m=int(r.integers(1, min(A, 16) + 1)), seed=400 + seed,
                legal=[None, "random"][int(r.integers(0, 2))] if A == 9 else None, noise_w=[0.25, None][int(r.integers(0, 2))],
                discount=float(r.choice([0.997, 1.0])), zero=bool(r.random() < 0.1))
    if case["legal"] is None and seed % 3 == 1:
        case["legal"] = "random"      # ragged legal lists on a third of the seeds (full lists otherwise)
    c = gd.make_inputs(case)
    dev = gd.run_tree(gdev(c), c)
    ora = gd.run_tree(octree.gmz_tree, c, roots_kwargs=dict(action_space_size=c["A"], max_simulations=c["S"]))
    gd.assert_same(ora, dev, repr(case))
