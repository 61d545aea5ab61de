"""CPU: the device math header (lightzero_amd/csrc/lz_math.h) must reproduce the host libm's
expf/logf -- the functions the reference tree calls (cnode.cpp:134,:776) -- bit for bit.
Exhaustive over all 2^32 binary32 inputs (about 15 s on 8 cores)."""
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def test_lz_math_matches_libm_exhaustively():
    exe = os.path.join(tempfile.mkdtemp(prefix="lzmath_"), "lz_math_check")
    subprocess.run(["gcc", "-O2", "-std=gnu11", "-mfma", "-ffp-contract=off", "-fopenmp", "-o", exe,
                    os.path.join(HERE, "lz_math_check.c"), "-lm"], check=True)
    stride = os.environ.get("LZ_MATH_STRIDE", "1")
    r = subprocess.run([exe, stride], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "expf mismatches 0" in r.stdout and "logf mismatches 0" in r.stdout
