"""Measured floating-point parity, kept as data.  Every GPU numerics test reports the worst difference it saw per tensor class
through ``record``; the numbers are merged into one JSON per kernel variant (the set of LZ_* switches in the environment)
under gpurun_out/parity/ on the GPU box, and tools/refresh_profiles.py commits their union as profiles/rNN_parity.json.

Bounds (``BOUNDS``): north_star asks for 1e-5.  It holds -- and is asserted -- for every tensor the network produces BEFORE
the inverse scalar transform (latent state, LSTM h / c, policy logits, support-wide value / reward logits), measured as
|d| / (1 + |x|).  The scalars AFTER h^-1 (value, value prefix, reward) cannot meet 1e-5 in ANY fp32 implementation whose
summation order differs from torch's: the reference formula (scaling_transform.py:88-91) subtracts 1 from sqrt(...) ~ 1.004 and
divides by 2e-3, so its own output moves in steps of ~1.3e-4 (1 + |x|); their bound is 3e-4 (1 + |x|) and
the pre-transform logits they are computed from are held to 1e-5."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# tensor class -> asserted bound on |d| / (1 + |x|)
BOUNDS = {
    "latent": 1e-5, "h": 1e-5, "c": 1e-5, "hc": 1e-5, "policy": 1e-5, "logits": 1e-5,
    # split heads (the production path of 49 of 50 simulations): support-wide logits and the pre-transform expectation softmax . support
    "logits_split": 1e-5, "expect": 1e-5, "expect_split": 1e-5,
    "scalar": 3e-4, "value": 3e-4, "value_prefix": 3e-4, "reward": 3e-4,
}


def variant():
    knobs = sorted(k for k in os.environ if k.startswith("LZ_") and k not in ("LZ_PARITY_OUT", "LZ_TEST_TORCH_THREADS", "LZ_MI355_LIB", "LZ_REFERENCE_ROOT"))
    return "+".join("%s=%s" % (k, os.environ[k]) for k in knobs) or "default"


def _path():
    out = os.environ.get("LZ_PARITY_OUT")
    if out:
        return out
    d = os.path.join(ROOT, "gpurun_out", "parity")
    os.makedirs(d, exist_ok=True)
    return os.path.join(d, "parity_%s.json" % variant().replace("=", "-").replace("+", "_"))


def record(test, worst, extra=None, bounds=None):
    """worst: {tensor class: worst |d| / (1 + |x|)}; merged (max) into this variant's JSON.  ``bounds``: a test that asserts other bounds
    than ``BOUNDS`` (the randomised model sweep ties them to what fp32 loses against binary64 on each network) records them with its
    entry; such entries are summarised separately (tools/refresh_profiles.py)"""
    import fcntl
    path = _path()
    with open(path + ".lock", "w") as lock:     # pytest-xdist workers share the variant's file: read-modify-write under a lock
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            data = json.load(open(path))
        except Exception:
            data = {"variant": variant(), "unit": "max |device - reference| / (1 + |reference|)", "bounds": BOUNDS, "tests": {}}
        ent = data["tests"].setdefault(test, {})
        for k, v in worst.items():
            ent[k] = max(float(v), float(ent.get(k, 0.0)))
        if extra:
            ent.update(extra)
        if bounds:
            ent["bounds"] = {k: float(v) for k, v in bounds.items()}
        with open(path, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)


def check(test, worst, extra=None, bounds=None):
    """record, then assert every class against its bound"""
    record(test, worst, extra, bounds)
    b = dict(BOUNDS)
    b.update(bounds or {})
    bad = {k: (float(v), b[k]) for k, v in worst.items() if not float(v) < b[k]}
    assert not bad, "%s: worst |d|/(1+|x|) above the bound (measured, bound): %s   all: %s" % (test, bad, worst)


def check_vs_truth(test, dev_vs_ref, dev_vs_truth, ref_vs_truth, extra=None):
    """The same gate on a network whose own fp32 evaluation is not exact to 1e-5 (deep boards without an LSTM reset: the latent's magnitude
    grows with every unrolled step).  ``dev_vs_ref``: device against the torch fp32 reference; ``dev_vs_truth`` / ``ref_vs_truth``: both
    against a binary64 evaluation of the same network on the same teacher-forced inputs (the exact value up to 1e-16).  Asserted per tensor
    class: the device is within the class bound of the EXACT value, and within bound + the reference's own distance from the exact value
    of the reference (triangle inequality: two fp32 evaluations that are each d from the truth may be 2 d apart).  All three are recorded."""
    ex = dict(extra or {})
    ex["device_vs_binary64"] = {k: float(v) for k, v in dev_vs_truth.items()}
    ex["torch_fp32_vs_binary64"] = {k: float(v) for k, v in ref_vs_truth.items()}
    bounds = {k: BOUNDS[k] + float(ref_vs_truth.get(k, 0.0)) for k in dev_vs_ref}
    record(test, dev_vs_ref, ex, bounds if any(float(ref_vs_truth.get(k, 0.0)) > 0 for k in dev_vs_ref) else None)
    bad = {k: (float(v), BOUNDS[k]) for k, v in dev_vs_truth.items() if not float(v) < BOUNDS[k]}
    assert not bad, "%s: the device is further than the bound from the binary64 evaluation (measured, bound): %s" % (test, bad)
    bad = {k: (float(v), bounds[k]) for k, v in dev_vs_ref.items() if not float(v) < bounds[k]}
    assert not bad, "%s: worst |d|/(1+|x|) above bound + the reference's own distance from binary64 (measured, bound): %s" % (test, bad)
