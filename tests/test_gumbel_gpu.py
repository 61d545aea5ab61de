"""GPU parity (through the C ABI): the Gumbel MuZero tree kernels vs the CPU oracle and the golden vectors generated from the
reference's own compiled gmz_tree -- bit-exact records, visit counts, root values, improved policies, completed values."""
import os

import numpy as np
import pytest

import gumbel_driver as gd

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _dev_mod(c):
    from lightzero_amd.mcts.ctree.ctree_gumbel_muzero import gmz_tree as m

    def mk(n, legal, **kw):
        return m.Roots(n, legal, action_space_size=c["A"], max_simulations=c["S"])
    return type("M", (), dict(Roots=staticmethod(mk), MinMaxStatsList=m.MinMaxStatsList, ResultsWrapper=m.ResultsWrapper,
                              batch_traverse=staticmethod(m.batch_traverse), batch_back_propagate=staticmethod(m.batch_back_propagate)))


@pytest.mark.parametrize("name", sorted(gd.CASES))
def test_device_gumbel_tree_matches_oracle_and_golden(name):
    from oracle import ctree as octree
    c = gd.make_inputs(gd.CASES[name])
    dev = gd.run_tree(_dev_mod(c), c)
    ora = gd.run_tree(octree.gmz_tree, c, roots_kwargs=dict(action_space_size=c["A"], max_simulations=c["S"]))
    gd.assert_same(ora, dev, name)
    g = np.load(os.path.join(GOLD, "gumbel_%s.npz" % name))
    assert np.array_equal(dev["records"], g["records"])
    assert np.array_equal(dev["policies"].view(np.uint32), g["policies"].view(np.uint32))
