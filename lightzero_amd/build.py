"""Builds liblz_mi355.so (HIP kernels + C ABI) for gfx950, in-tree, with hipcc.

    python -m lightzero_amd.build [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the working tree.
"""
import os
import re
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "liblz_mi355.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# translation unit -> extra flags.  The tree kernels reproduce the reference's float arithmetic
# exactly, so they are built without FMA contraction (see lz_tree.hip header).
UNITS = [
    ("lz_tree.hip", ["-ffp-contract=off"]),
    ("lz_tree_sampled.hip", ["-ffp-contract=off"]),
    ("lz_tree_wide.hip", ["-ffp-contract=off"]),
    ("lz_capi.hip", []),
    ("lz_nn.hip", []),
    ("lz_chain_s3g.hip", []),
    ("lz_dense.hip", []),
    ("lz_mlp.hip", []),
    ("lz_search.hip", []),
]


def csrc_digest():
    """sha256 over the sources the library is built from (csrc/*.hip, csrc/*.h, include/lz_mi355.h): what a committed profile or
    parity record was measured on.  profiles/rNN_manifest.json carries it; bench.py trusts the committed rocprofv3 numbers only when
    it matches the tree it runs from, and tests/test_profiles_cpu.py fails when a kernel source changed after the last profile."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    for f in files:
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(os.path.dirname(PKG), "include", "lz_mi355.h"), "rb").read())
    return h.hexdigest()


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


DBG_LIB = os.path.join(PKG, "liblz_mi355_dbg.so")


def build(force=False, verbose=False, debug_knobs=False):
    """debug_knobs=True builds liblz_mi355_dbg.so instead: the same sources with -DLZ_DEBUG_KNOBS, i.e. WITH the skip-work timing
    switches (LZ_DEBUG_SKIP, LZ_DEBUG_CHAIN_LAYERS, LZ_DEBUG_CHAIN_TS, LZ_DEBUG_LSTM_ROWS) that the release library does not
    contain.  Only tools/ load it (LZ_MI355_LIB=<path>)."""
    if not os.path.exists(HIPCC):
        raise RuntimeError("hipcc not found at %s" % HIPCC)
    LIB = DBG_LIB if debug_knobs else globals()["LIB"]
    osuffix = ".dbg.o" if debug_knobs else ".o"
    dflags = ["-DLZ_DEBUG_KNOBS"] if debug_knobs else []
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(PKG), "include", "lz_mi355.h"))
    objs = []
    relink = force or not os.path.exists(LIB)

    def compile_unit(src, extra, sp, op):
        # -Rpass-analysis=kernel-resource-usage: per-kernel registers / scratch as compiler remarks.  A kernel that falls
        # back to scratch memory (an array the optimiser could not keep in registers) costs a round trip per access on
        # these latency-bound kernels and was once a silent 5 % regression: refuse it.
        cmd = [HIPCC] + COMMON + dflags + extra + ["-Rpass-analysis=kernel-resource-usage", "-c", sp, "-o", op]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, stderr=subprocess.PIPE, universal_newlines=True)
        diag = "\n".join(l for l in res.stderr.splitlines() if "kernel-resource-usage" not in l and not re.match(r"^\s*(\d+ \|.*|\| +\^)\s*$", l))
        if diag.strip():
            sys.stderr.write(diag + "\n")
        if res.returncode != 0:
            raise subprocess.CalledProcessError(res.returncode, cmd)
        name = None
        for line in res.stderr.splitlines():
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                name = m.group(1)
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
            if m and int(m.group(1)) > 0 and not (debug_knobs and os.environ.get("LZ_BUILD_ALLOW_SCRATCH")):  # experiments with the timing instances only
                os.remove(op)
                raise RuntimeError("%s: kernel %s uses %s bytes of scratch per lane" % (src, name, m.group(1)))

    stale = []
    for src, extra in UNITS:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        op = sp[:-4] + osuffix
        if force or _newer(sp, op) or any(_newer(h, op) for h in hdrs):
            stale.append((src, extra, sp, op))
        objs.append(op)
    if stale:
        # the translation units are independent: compile them side by side (lz_nn.hip alone is two thirds of a full build)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(stale), os.cpu_count() or 1)) as ex:
            for f in [ex.submit(compile_unit, *u) for u in stale]:
                f.result()
        relink = True
    if relink or any(_newer(o, LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, debug_knobs="--debug-knobs" in sys.argv))
