"""GPU: the alternative kernel paths behind the LZ_* switches of INTEGRATION.md §2d (direct-form chain / tower, 4-wave Winograd chain, VALU
heads, unsplit LSTM staging, the fp32-matrix (Winograd) tower instead of the split-bf16 one, the level-by-level tree walk, the pipelined 32-row LSTM kernel, separate tree launch) must meet the same reference-module goldens as the default path.  The switches are read
once per process, so every variant runs tests/test_nn_golden_gpu.py in a process of its own."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = ["LZ_CHAIN_DIRECT", "LZ_CHAIN_W4", "LZ_CONV_DIRECT", "LZ_CONV_NO_SPLIT", "LZ_CHAIN_NO_SPLIT", "LZ_TRAVERSE_SERIAL", "LZ_CONV_FIRST_VALU", "LZ_HEADS_VALU", "LZ_HEADS_LAUNCH", "LZ_LSTM_NOSPLIT", "LZ_NO_TREE_FUSE", "LZ_NO_GRAPH", "LZ_LSTM3"]


@pytest.mark.gpu
@pytest.mark.parametrize("knob", VARIANTS)
def test_nn_goldens_hold_on_the_alternative_path(knob):
    env = dict(os.environ)
    env[knob] = "1"
    # the goldens (teacher-forced single steps) and the teacher-forced check of every simulation of a captured search at B = 67 / 256
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_nn_golden_gpu.py"),
                        os.path.join(ROOT, "tests", "test_nn_gpu.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "%s=1:\n%s\n%s" % (knob, r.stdout[-3000:], r.stderr[-2000:])
    assert " passed" in r.stdout


@pytest.mark.gpu
def test_one_launch_per_simulation_replays_exactly_and_reports_instead_of_hanging():
    """LZ_SIM_ONE_LAUNCH=1 (k_sim_fused: the LSTM launch of simulation s - 1 and the tree-fused chain launch of simulation s as two phases of one
    launch, 16-root groups handing the head partials over through one XCD's L2 -- VERDICT r5 #3's hand-off; measured SLOWER than two launches,
    profiles/r06_one_launch_ab.txt, and kept opt-in): the production launch sequence at BASELINE configs[1] size still replays exactly through
    the reference's compiled ctree, two runs are bit-identical, and a search whose batch is not 128 | 256 roots takes the two-launch path."""
    env = dict(os.environ)
    env["LZ_SIM_ONE_LAUNCH"] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_exact_replay_gpu.py"),
                        os.path.join(ROOT, "tests", "test_determinism_gpu.py"), os.path.join(ROOT, "tests", "test_end_to_end_gpu.py"),
                        "-x", "-q", "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "LZ_SIM_ONE_LAUNCH=1:\n%s\n%s" % (r.stdout[-3000:], r.stderr[-2000:])
    assert " passed" in r.stdout
