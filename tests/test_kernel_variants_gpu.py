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
