"""CPU: pins oracle/ctree_oracle.c against (a) the reference's own compiled ctree (oracle/_ref/det,
only where /root/reference exists) and (b) the committed golden vectors generated from it."""
import os

import numpy as np
import pytest

import tree_driver as td
from oracle import build_ref, ctree as octree

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _mod(variant):
    return octree.ez_tree if variant == "ez" else octree.mz_tree


@pytest.mark.parametrize("name", sorted(td.CASES))
def test_oracle_matches_compiled_reference(name):
    if not build_ref.build():
        pytest.skip("reference sources not present (GPU box)")
    ez_ref, mz_ref = build_ref.load("det")
    c = td.make_inputs(td.CASES[name])
    ref_mod = ez_ref if c["variant"] == "ez" else mz_ref
    kw = dict(traverse_kwargs=dict(deterministic=True)) if c["variant"] == "mz" else {}
    ref = td.run_tree(ref_mod, c, **kw)
    ora = td.run_tree(_mod(c["variant"]), c, roots_kwargs=dict(action_space_size=c["A"], max_simulations=c["S"]))
    td.assert_same(ref, ora, name)


@pytest.mark.parametrize("name", sorted(td.CASES))
def test_oracle_matches_golden(name):
    path = os.path.join(GOLD, "tree_%s.npz" % name)
    g = np.load(path)
    c = td.make_inputs(td.CASES[name])
    ora = td.run_tree(_mod(c["variant"]), c, roots_kwargs=dict(action_space_size=c["A"], max_simulations=c["S"]))
    assert np.array_equal(ora["records"], g["records"])
    dist = np.full((c["B"], c["A"]), -1, np.int32)
    for i, d in enumerate(ora["distributions"]):
        dist[i, :len(d)] = d
    assert np.array_equal(dist, g["distributions"])
    assert np.array_equal(ora["values"].view(np.uint32), g["values"].view(np.uint32))


def test_reference_known_answer_deterministic_first_action():
    # lzero/mcts/tests/test_muzero_ctree_deterministic.py:4-25 -- the reference's only value-level pin
    mz = octree.mz_tree
    roots = mz.Roots(1, [[0, 1, 2]], action_space_size=3)
    roots.prepare_no_noise([0.0], [[0.0, 0.0, 0.0]], [-1])
    mm = mz.MinMaxStatsList(1)
    mm.set_delta(0.01)
    sel = []
    for _ in range(5):
        res = mz.ResultsWrapper(1)
        sel.append(mz.batch_traverse(roots, 19652, 1.25, 0.997, mm, res, [-1], deterministic=True)[2][0])
    assert sel == [0] * 5


def test_reference_stochastic_tie_breaking_mode():
    # lzero/mcts/tests/test_muzero_ctree_deterministic.py:28-48
    mz = octree.mz_tree
    roots = mz.Roots(1, [[0, 1, 2]], action_space_size=3)
    roots.set_tiebreak(1)
    roots.prepare_no_noise([0.0], [[0.0, 0.0, 0.0]], [-1])
    mm = mz.MinMaxStatsList(1)
    mm.set_delta(0.01)
    sel = []
    for _ in range(30):
        res = mz.ResultsWrapper(1)
        sel.append(mz.batch_traverse(roots, 19652, 1.25, 0.997, mm, res, [-1])[2][0])
    assert len(set(sel)) > 1


# ---- ReZero search_with_reuse ---------------------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(td.REUSE_CASES))
def test_oracle_reuse_matches_compiled_reference(name):
    if not build_ref.build():
        pytest.skip("reference sources not present (GPU box)")
    ez_ref, mz_ref = build_ref.load("det")
    c = td.make_reuse_inputs(td.REUSE_CASES[name])
    ref = td.run_tree_reuse(ez_ref if c["variant"] == "ez" else mz_ref, c)
    ora = td.run_tree_reuse(_mod(c["variant"]), c, roots_kwargs=dict(action_space_size=c["A"], max_simulations=c["S"]))
    td.assert_same(ref, ora, name)
    assert ref["inferences"] == ora["inferences"] and ref["inferences"] < c["B"] * c["S"]  # some roots did skip inference


@pytest.mark.parametrize("name", sorted(td.REUSE_CASES))
def test_oracle_reuse_matches_golden(name):
    g = np.load(os.path.join(GOLD, "tree_%s.npz" % name))
    c = td.make_reuse_inputs(td.REUSE_CASES[name])
    ora = td.run_tree_reuse(_mod(c["variant"]), c, roots_kwargs=dict(action_space_size=c["A"], max_simulations=c["S"]))
    assert np.array_equal(ora["records"], g["records"])
    dist = np.full((c["B"], c["A"]), -1, np.int32)
    for i, d in enumerate(ora["distributions"]):
        dist[i, :len(d)] = d
    assert np.array_equal(dist, g["distributions"])
    assert np.array_equal(ora["values"].view(np.uint32), g["values"].view(np.uint32))
