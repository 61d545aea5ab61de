"""CPU: pins oracle/ctree_gumbel_oracle.c against the reference's own compiled Gumbel MuZero ctree (oracle/_ref/stock, only where
/root/reference exists) and against the committed golden vectors generated from it."""
import os

import numpy as np
import pytest

import gumbel_driver as gd
from oracle import build_ref, ctree as octree

GOLD = os.path.join(os.path.dirname(__file__), "golden")
KW = lambda c: dict(action_space_size=c["A"], max_simulations=c["S"])  # noqa: E731


@pytest.mark.parametrize("name", sorted(gd.CASES))
def test_gumbel_oracle_matches_compiled_reference(name):
    if not build_ref.build():
        pytest.skip("reference sources not present (GPU box)")
    ref_mod = build_ref.load_gumbel()
    c = gd.make_inputs(gd.CASES[name])
    ref = gd.run_tree(ref_mod, c)
    ora = gd.run_tree(octree.gmz_tree, c, roots_kwargs=KW(c))
    gd.assert_same(ref, ora, name)


@pytest.mark.parametrize("name", sorted(gd.CASES))
def test_gumbel_oracle_matches_golden(name):
    g = np.load(os.path.join(GOLD, "gumbel_%s.npz" % name))
    c = gd.make_inputs(gd.CASES[name])
    ora = gd.run_tree(octree.gmz_tree, c, roots_kwargs=KW(c))
    gd.assert_same({k: g[k].astype(ora[k].dtype) if k == "records" else g[k] for k in g.files}, ora, name)


def test_gumbel_helpers_match_reference():
    if not build_ref.build():
        pytest.skip("reference sources not present (GPU box)")
    ref = build_ref.load_gumbel()
    L = octree.glib()
    out = np.zeros(40, np.float32)
    L.gtree_generate_gumbel(10.0, 0.0, 40, out)
    assert np.array_equal(out.view(np.uint32), np.asarray(ref.pgenerate_gumbel(10.0, 0.0, 40), np.float32).view(np.uint32))
    for m in (1, 2, 4, 8, 16):
        for n in (5, 25, 50, 200):
            seq = np.zeros(n, np.int32)
            L.gtree_considered_visits(m, n, seq)
            assert seq.tolist() == list(ref.pget_sequence_of_considered_visits(m, n))
