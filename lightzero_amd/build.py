"""Builds liblz_mi355.so (HIP kernels + C ABI) for gfx950, in-tree, with hipcc.

    python -m lightzero_amd.build [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the working tree.
"""
import os
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
    ("lz_capi.hip", []),
    ("lz_nn.hip", []),
    ("lz_dense.hip", []),
    ("lz_mlp.hip", []),
    ("lz_search.hip", []),
]


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=False):
    if not os.path.exists(HIPCC):
        raise RuntimeError("hipcc not found at %s" % HIPCC)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(PKG), "include", "lz_mi355.h"))
    objs = []
    relink = force or not os.path.exists(LIB)
    for src, extra in UNITS:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        op = sp[:-4] + ".o"
        if force or _newer(sp, op) or any(_newer(h, op) for h in hdrs):
            cmd = [HIPCC] + COMMON + extra + ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
            relink = True
        objs.append(op)
    if relink or any(_newer(o, LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
