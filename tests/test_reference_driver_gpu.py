"""GPU: the recorded runs of the reference's OWN search drivers (tests/golden/driver_*.npz, made by tests/golden/make_golden_driver.py
from /root/reference/lzero/mcts/tree_search/mcts_ctree.py on the reference's compiled ctree) replayed through lightzero_amd's
reference-named drivers with a foreign model -- the reference's loop (mcts_ctree.py:782-876 / 300-368) around the HBM trees' batch_traverse
/ batch_backpropagate -- must make the reference's selections in every simulation and end at the reference's visit counts and
BIT-EQUAL root values.  (swapping the ctree module under the reference driver.)"""
import os
import sys

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
import make_golden_driver as mgd  # noqa: E402
from test_reference_driver_cpu import replay_golden  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(mgd.CASES))
def test_device_tree_behind_the_reference_loop_replays_the_reference_run(name):
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree
    from lightzero_amd.mcts.tree_search.mcts_ctree import EfficientZeroMCTSCtree, MuZeroMCTSCtree
    g = np.load(os.path.join(GOLD, name + ".npz"))
    case = mgd.CASES[name]
    ez = case["family"] == "ez"

    class Tree(object):  # the module surface with deterministic ties (the golden was made with rand() -> 0)
        @staticmethod
        def Roots(n, legal, **kw):
            r = (ez_tree if ez else mz_tree).Roots(n, legal, **kw)
            r.set_tiebreak(0)
            return r

    def search(roots, model, handle, lat0, hc0, to_play, cfg):
        mcts = EfficientZeroMCTSCtree(cfg) if ez else MuZeroMCTSCtree(cfg)
        # the recorded post-transform scalars stand in for the handles' torch arithmetic (machine-dependent in its last bits)
        mcts.value_inverse_scalar_transform_handle = mcts.reward_inverse_scalar_transform_handle = handle
        if ez:
            mcts.search(roots, model, lat0, hc0, to_play)
        else:
            mcts.search(roots, model, lat0, to_play)
    dist, values = replay_golden(case, g, Tree, roots_kwargs=dict(action_space_size=case["A"], max_simulations=case["S"]), search=search)
    assert np.array_equal(dist, g["distributions"])
    assert np.array_equal(values.view(np.uint32), g["values"].view(np.uint32))
