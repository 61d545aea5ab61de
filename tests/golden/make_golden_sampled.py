"""Generates tests/golden/sampled_<case>.npz from the reference's OWN compiled Sampled-EfficientZero tree
(oracle/_ref/det: rand() -> 0, system_clock::now() -> settable counter).  Needs /root/reference:

    python tests/golden/make_golden_sampled.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import sampled_driver as sd  # noqa: E402
from oracle import build_ref  # noqa: E402

CLOCK0 = 123456789
assert build_ref.build(), "reference not present"
ezs, h = build_ref.load_sampled("det")
for name in sorted(sd.CASES):
    c = sd.make_inputs(sd.CASES[name])
    h.oracle_set_clock(CLOCK0)
    out = sd.run_tree(ezs, c, lambda: ezs.Roots(c["B"], [[-1] * 5 for _ in range(c["B"])], c.get("A") or c["D"], c["K"], not c.get("A")))
    np.savez_compressed(os.path.join(HERE, "sampled_%s.npz" % name), records=out["records"].astype(np.int16),
                        distributions=out["distributions"].astype(np.int16), values=out["values"],
                        root_actions=out["root_actions"], last_actions=out["last_actions"])
    print(name, "ok; max depth", out["records"][:, :, 2].max())
