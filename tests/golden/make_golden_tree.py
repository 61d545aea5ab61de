"""Generates tests/golden/tree_<case>.npz from the reference's OWN compiled ctree (oracle/_ref/det =
reference sources + rand()->0).  Run in the build container (needs /root/reference):

    python tests/golden/make_golden_tree.py [case-name-prefix ...]      (no argument: every case)

Inputs are re-derived from the seeds in tests/tree_driver.py::CASES, so only outputs are stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import tree_driver as td  # noqa: E402
from oracle import build_ref  # noqa: E402

assert build_ref.build(), "reference not present"
ez_ref, mz_ref = build_ref.load("det")
ONLY = tuple(sys.argv[1:])
for name in sorted(td.CASES):
    if ONLY and not name.startswith(ONLY):
        continue
    c = td.make_inputs(td.CASES[name])
    mod = ez_ref if c["variant"] == "ez" else mz_ref
    kw = dict(traverse_kwargs=dict(deterministic=True)) if c["variant"] == "mz" else {}
    out = td.run_tree(mod, c, **kw)
    dist = np.full((c["B"], c["A"]), -1, np.int32)
    for i, d in enumerate(out["distributions"]):
        dist[i, :len(d)] = d
    rec = out["records"]
    dt = np.int8 if rec.max() < 127 and rec.min() >= -128 else (np.int16 if rec.max() < 32767 else np.int32)
    np.savez_compressed(os.path.join(HERE, "tree_%s.npz" % name), records=rec.astype(dt), distributions=dist,
                        values=out["values"])
    print(name, "ok", rec.shape, "max depth", rec[:, :, 3].max())

# ReZero search_with_reuse
for name in sorted(td.REUSE_CASES):
    if ONLY and not name.startswith(ONLY):
        continue
    c = td.make_reuse_inputs(td.REUSE_CASES[name])
    out = td.run_tree_reuse(ez_ref if c["variant"] == "ez" else mz_ref, c)
    dist = np.full((c["B"], c["A"]), -1, np.int32)
    for i, d in enumerate(out["distributions"]):
        dist[i, :len(d)] = d
    rec = out["records"]
    dt = np.int8 if rec.max() < 127 and rec.min() >= -128 else (np.int16 if rec.max() < 32767 else np.int32)
    np.savez_compressed(os.path.join(HERE, "tree_%s.npz" % name), records=rec.astype(dt), distributions=dist,
                        values=out["values"])
    print(name, "ok", rec.shape, "inferences", out["inferences"], "of", c["B"] * c["S"])
