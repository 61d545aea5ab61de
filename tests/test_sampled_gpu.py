"""GPU: Sampled-EfficientZero tree kernels (continuous actions) through the C ABI.
* parity: with the oracle's draws injected (``roots.given``), every observable is bit-identical to the CPU oracle
  (which is itself pinned to the reference's compiled module) and to the committed goldens;
* device-side sampling: the reference seeds its generator from the wall clock, so only the DISTRIBUTION is defined:
  tanh(N(mu, sigma)) moments and a Kolmogorov-Smirnov check, plus tree invariants."""
import os

import numpy as np
import pytest

import sampled_driver as sd

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
CLOCK0 = 123456789


@pytest.mark.parametrize("name", sorted(sd.CASES))
def test_device_sampled_tree_matches_oracle_with_injected_draws(name):
    from oracle import ctree as octree
    from lightzero_amd.mcts.ctree.ctree_sampled_efficientzero import ezs_tree
    c = sd.make_inputs(sd.CASES[name])
    draws = {}

    def mk_o():
        r = octree.ezs_tree.Roots(c["B"], [[-1] * 5] * c["B"], c["D"], c["K"], True, max_simulations=c["S"])
        r.set_clock(CLOCK0)
        return r
    ora = sd.run_tree(octree.ezs_tree, c, mk_o, after_expand=lambda r, e: draws.__setitem__(e, np.asarray(r.get_sampled_actions(e), np.float32)))

    def mk_d():
        r = ezs_tree.Roots(c["B"], [[-1] * 5] * c["B"], c["D"], c["K"], True, max_simulations=c["S"])
        r.set_tiebreak(0)
        return r
    dev = sd.run_tree(ezs_tree, c, mk_d, before_expand=lambda r, e: setattr(r, "given", draws[e]))
    sd.assert_same(ora, dev, name)
    g = np.load(os.path.join(GOLD, "sampled_%s.npz" % name))
    assert np.array_equal(dev["records"], g["records"]) and np.array_equal(dev["distributions"], g["distributions"])
    assert np.array_equal(dev["values"].view(np.uint32), g["values"].view(np.uint32))


def test_device_side_sampling_distribution_and_invariants():
    from scipy import stats
    from lightzero_amd.mcts.ctree.ctree_sampled_efficientzero import ezs_tree
    B, D, K, S = 256, 1, 20, 50
    rng = np.random.default_rng(0)
    mu, sigma = 0.3, 0.6
    pol = np.tile(np.array([[mu, sigma]], np.float32), (B, 1))
    roots = ezs_tree.Roots(B, [[-1] * 5] * B, D, K, True, max_simulations=S)
    roots.prepare_no_noise([0.0] * B, pol.tolist(), [-1] * B)
    acts = np.asarray(roots.get_sampled_actions(), np.float64).reshape(-1)
    assert (np.abs(acts) < 1).all()
    z = (np.arctanh(np.clip(acts, -0.999999, 0.999999)) - mu) / sigma  # should be N(0,1)
    ks = stats.kstest(z, "norm")
    assert ks.pvalue > 1e-3, ks
    assert abs(z.mean()) < 0.05 and abs(z.std() - 1) < 0.05
    mm = ezs_tree.MinMaxStatsList(B)
    mm.set_delta(0.01)
    for s in range(S):
        res = ezs_tree.ResultsWrapper(B)
        ix, iy, la, vtp = ezs_tree.batch_traverse(roots, 19652, 1.25, 0.997, mm, res, [-1] * B, True)
        sl = res.get_search_len()
        assert max(ix) <= s and (np.abs(np.asarray(la)) < 1).all()
        ezs_tree.batch_backpropagate(s + 1, 0.997, (0.1 * rng.standard_normal(B)).astype(np.float32).tolist(),
                                     rng.standard_normal(B).astype(np.float32).tolist(), pol.tolist(), mm, res,
                                     [int(l % 5 == 0) for l in sl], vtp)
    dist = np.asarray(roots.get_distributions())
    assert dist.shape == (B, K) and (dist.sum(1) == S).all()
    # two roots must not share a random stream
    ra = np.asarray(roots.get_sampled_actions())
    assert not np.array_equal(ra[0], ra[1])
