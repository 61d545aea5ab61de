"""GPU (1-GPU box): the multi-rank path of bench.py end to end -- `bench.py --gpus 2` re-executes itself under torch.distributed.run, two
ranks share the device, the row collective falls back to gloo on host copies (flagged as such: never a headline number).  Checks the
JSON contract, per-rank values, that every rank's pooled rows equal the ranks' own rows (--check-gather), the strong-scaling mode
(--total-envs with uneven blocks) and the weight refresh inside the loop (--refresh-every: broadcast + in-place re-ingest)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(*extra, gpus="2", env_extra=None):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **(env_extra or {}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", gpus, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--sustain-s", "0",
                        "--check-gather"] + list(extra), cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_two_ranks_weak_scaling_rows_pooled_and_weights_refreshed():
    d = _run("--refresh-every", "2")
    c = d["config"]
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and c["envs_per_rank"] == [256, 256] and c["total_envs"] == 512
    assert c["gather_check"] == "ok" and c["weight_refresh_every"] == 2
    assert c["collective_backend"] in ("gloo", "nccl") and c["rccl_ranks"] in (0, 2)
    assert len(c["per_rank_env_steps_per_s"]) == 2 and all(v > 0 for v in c["per_rank_env_steps_per_s"])
    assert d["value"] > 0 and abs(d["value"] - 512 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    assert d["roofline"]["frac"] is not None and c["debug_knobs"] == []


def test_two_ranks_strong_scaling_uneven_blocks():
    d = _run("--total-envs", "130")
    c = d["config"]
    assert d["scaling"] == "strong" and c["envs_per_rank"] == [65, 65] and c["total_envs"] == 130 and c["gather_check"] == "ok"
    d = _run("--total-envs", "129")
    c = d["config"]
    assert c["envs_per_rank"] == [65, 64] and c["gather_check"] == "ok" and d["value"] > 0


@pytest.mark.parametrize("fence", ["tower", "none"])
def test_rccl_branch_with_one_rank(fence):
    """The RCCL branch of the collective path on the 1-GPU box (LZ_FORCE_COLLECTIVE=1: a one-rank process group): device-side
    asynchronous all-gather into the double-buffered pool, the search ordered behind the previous step's collective on the engine's
    stream (--gather-fence tower), weight broadcast, the pooled rows verified."""
    d = _run("--refresh-every", "2", "--gather-fence", fence, gpus="1", env_extra={"LZ_FORCE_COLLECTIVE": "1"})
    c = d["config"]
    assert d["n_gpus"] == 1 and c["collective_backend"] == "nccl" and c["rccl_ranks"] == 1
    assert c["all_gather_overlapped"] is True and c["all_gather_fence"] == fence and c["gather_check"] == "ok"
    assert d["value"] > 0
