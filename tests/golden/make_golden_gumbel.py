"""Generates tests/golden/gumbel_<case>.npz from the reference's OWN compiled Gumbel MuZero ctree (oracle/_ref/stock).
Run in the build container (needs /root/reference):  python tests/golden/make_golden_gumbel.py [case ...]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import gumbel_driver as gd  # noqa: E402
from oracle import build_ref  # noqa: E402

assert build_ref.build(), "reference not present"
mod = build_ref.load_gumbel()
for name in sorted(gd.CASES):
    if sys.argv[1:] and name not in sys.argv[1:]:
        continue
    c = gd.make_inputs(gd.CASES[name])
    out = gd.run_tree(mod, c)
    np.savez_compressed(os.path.join(HERE, "gumbel_%s.npz" % name), records=out["records"].astype(np.int16),
                        distributions=out["distributions"], values=out["values"], policies=out["policies"],
                        children_values=out["children_values"])
    print(name, "ok", out["records"].shape, "max depth", out["records"][:, :, 3].max())
