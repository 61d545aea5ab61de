"""GPU: the single-GPU bench line -- the contract's fields, the roofline / cpu_baseline objects, the separately reported fast-mode arm, and the
`--fast` arm's own line (never the parity-mode headline: another metric string, dtype bf16, no committed-profile figures)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(*extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--sustain-s", "0"] + list(extra),
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_default_line_is_parity_mode_and_carries_the_fast_arm_separately():
    d = _bench()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["dtype"] == "f32" and d["n_gpus"] == 1 and d["steps"] == 4 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "FAST" not in d["metric"] and "mode" not in d["config"]
    assert abs(d["value"] - 256 * 4 / (d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    # parity mode since round 5: split-bf16 products, priced on the bf16 dense peak / 6 (the fp32-matrix-pipe figure rides along)
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.15 < rf["frac"] < 1.0
    assert "k_chain_s3" in rf["kernel"] and abs(rf["peak"] - 2500.0 / 6) < 1e-6 and rf["frac_vs_fp32_matrix_peak"] > rf["frac"]
    assert abs(rf["executed_bf16_tflops"] - rf["achieved"] * 8.0) < 1e-6 * rf["achieved"]
    assert "this run" in rf["clock"] and rf["per_simulation_us"] > 0
    fm = d["fast_mode"]
    assert "error" not in fm and fm["env_steps_per_s"] > 0 and "bf16" in fm["dtype"] and "statistical parity only" in fm["note"]
    if not os.environ.get("PYTEST_XDIST_WORKER"):   # a comparison of two timings: only with the GPU to itself (pytest -n shares it between workers)
        assert fm["env_steps_per_s"] > d["value"]


def test_fast_line_is_labelled_and_keeps_the_parity_profile_out():
    d = _bench("--fast")
    assert "FAST MODE" in d["metric"] and d["dtype"] == "bf16" and "statistical parity only" in d["config"]["mode"]
    rf = d["roofline"]
    assert rf["traffic"] is None and rf["frac_profile"] is None and rf["peak"] > 1000 and "k_chain_b" in rf["kernel"]
    assert "fast_mode" not in d and d["value"] > 0
