#!/usr/bin/env python3
"""Fold the measured-parity records of ad-hoc wider sweeps (LZ_FUZZ_SEED_OFFSET=n runs of tests/test_nn_fuzz_gpu.py on the GPU box: one
gpurun_out/parity/parity_LZ_FUZZ_SEED_OFFSET-n.json each, written by tests/parity_record.py) into profiles/<tag>_parity_sweeps.json:
per offset the number of random networks, the worst |device - reference| / (1 + |reference|) per tensor class, the worst ratio to the bound each
network was held to (max(north_star's bound, 3 x torch fp32's own distance from binary64 on that network)), and every network above its bound.
The sweeps ran under pytest-xdist (-n 4) whose workers read-modify-write one file per offset: records lost to that race make `networks` a
lower bound of what ran (24 + 8 per full offset; skipped = refused configurations); a failing network is always in the pytest log as well.

    python tools/summarize_sweeps.py gpurun_out/parity r04
"""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(src, tag):
    out = {"_how": __doc__.strip().split("\n\n")[0].replace("\n", " "), "offsets": {}, "above_their_bound": []}
    n_all = 0
    for f in sorted(glob.glob(os.path.join(src, "parity_LZ_FUZZ_SEED_OFFSET-*.json")), key=lambda p: int(re.findall(r"-(\d+)\.json$", p)[0])):
        off = int(re.findall(r"-(\d+)\.json$", f)[0])
        d = json.load(open(f))
        worst, ratio, n = {}, 0.0, 0
        for t, ent in d["tests"].items():
            if not t.startswith("fuzz/"):
                continue
            n += 1
            b = dict(d["bounds"])
            b.update(ent.get("bounds") or {})
            for k, x in ent.items():
                if k in d["bounds"] and isinstance(x, float):
                    worst[k] = max(worst.get(k, 0.0), x)
                    ratio = max(ratio, x / b[k])
                    if not x < b[k]:
                        out["above_their_bound"].append({"offset": off, "test": t, "class": k, "measured": x, "bound": b[k]})
        out["offsets"][str(off)] = {"networks": n, "worst": worst, "worst_measured_over_bound": ratio}
        n_all += n
    out["networks"] = n_all
    out["unit"] = "max |device - reference| / (1 + |reference|)"
    path = os.path.join(ROOT, "profiles", "%s_parity_sweeps.json" % tag)
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print(path, n_all, "networks;", len(out["above_their_bound"]), "above their bound")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
