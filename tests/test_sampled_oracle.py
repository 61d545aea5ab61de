"""CPU: pins oracle/ctree_sampled_oracle.c (incl. its restatement of libstdc++'s minstd_rand0 / generate_canonical /
normal_distribution) against the reference's own compiled Sampled-EfficientZero tree (oracle/_ref/det: rand() -> 0,
system_clock::now() replaced by a settable counter) and against committed goldens generated from it."""
import os

import numpy as np
import pytest

import sampled_driver as sd
from oracle import build_ref, ctree as octree

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CLOCK0 = 123456789


def _run_oracle(c):
    def mk():
        r = octree.ezs_tree.Roots(c["B"], [[-1] * 5] * c["B"], c.get("A") or c["D"], c["K"], not c.get("A"), max_simulations=c["S"])
        r.set_clock(CLOCK0)
        return r
    return sd.run_tree(octree.ezs_tree, c, mk)


@pytest.mark.parametrize("name", [n for n in sorted(sd.CASES) if n != "sez_cfg5_b256"])
def test_sampled_oracle_matches_compiled_reference(name):
    if not build_ref.build():
        pytest.skip("reference sources not present (GPU box)")
    ezs, h = build_ref.load_sampled("det")
    c = sd.make_inputs(sd.CASES[name])
    h.oracle_set_clock(CLOCK0)
    ref = sd.run_tree(ezs, c, lambda: ezs.Roots(c["B"], [[-1] * 5 for _ in range(c["B"])], c.get("A") or c["D"], c["K"], not c.get("A")))
    ora = _run_oracle(c)
    sd.assert_same(ref, ora, name)
    if name == "sez_collide":  # the duplicate-key path must actually be exercised
        ra = ora["root_actions"]
        assert any(len({("%f" % v) for v in ra[b, :, 0]}) < c["K"] for b in range(c["B"]))


@pytest.mark.parametrize("name", sorted(sd.CASES))
def test_sampled_oracle_matches_golden(name):
    g = np.load(os.path.join(GOLD, "sampled_%s.npz" % name))
    c = sd.make_inputs(sd.CASES[name])
    ora = _run_oracle(c)
    assert np.array_equal(ora["records"], g["records"])
    assert np.array_equal(ora["distributions"], g["distributions"])
    assert np.array_equal(ora["values"].view(np.uint32), g["values"].view(np.uint32))
    assert np.array_equal(ora["root_actions"].view(np.uint32), g["root_actions"].view(np.uint32))
    assert np.array_equal(ora["last_actions"].view(np.uint32), g["last_actions"].view(np.uint32))
