"""TEST INFRASTRUCTURE ONLY -- builds the *reference's own* ctree extensions into oracle/_ref/.

Compiles, from the sources where they lie under /root/reference (nothing is copied into the
repository history; oracle/_ref/ is git-ignored), the reference Cython/C++ tree modules

    lzero/mcts/ctree/ctree_efficientzero/ez_tree.pyx  (+ lib/cnode.cpp, common_lib/cminimax.cpp)
    lzero/mcts/ctree/ctree_muzero/mz_tree.pyx         (+ lib/cnode.cpp, common_lib/cminimax.cpp)
    lzero/mcts/ctree/ctree_sampled_efficientzero/ezs_tree.pyx, lzero/mcts/ctree/ctree_gumbel_muzero/gmz_tree.pyx

exactly the way the reference's setup.py:67-96 does (cythonize, language=c++, -std=c++11), in
two flavours:

    oracle/_ref/stock/...  unmodified: ties broken by rand() after srand(tv_usec)  (cnode.cpp:691,901)
    oracle/_ref/det/...    same sources compiled with ``-include oracle_det.h`` which defines
                           rand() -> 0, so the tie list's first element (= first arg-max) is taken.
                           This is the deterministic parity oracle.

The staging copy lives in a temp dir (the .pxd is only found when the dotted extension name
matches the directory layout); only the built .so files land in oracle/_ref/.
Nothing on the product path imports this; see tests/ and bench.py's cpu_baseline leg.
"""
import os
import shutil
import subprocess
import sys
import tempfile

REF = os.environ.get("LZ_REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")

DET_HEADER = """\
// deterministic tie-break for the reference ctree: std headers FIRST (libstdc++'s <algorithm>
// itself refers to std::rand), then rand() -> 0.
#include <cstdlib>
#include <stdlib.h>
#include <algorithm>
#include <random>
#include <iostream>
#include <vector>
#include <map>
#include <stack>
#define rand() (0)
"""

CLOCK_CPP = """\
// deterministic replacement of std::chrono::system_clock::now() for the reference's sampled ctree (test oracle only)
#include <chrono>
#include <cstdint>
static uint64_t g_lz_clock = 1;
extern "C" void oracle_set_clock(uint64_t v) { g_lz_clock = v; }
extern "C" uint64_t oracle_get_clock(void) { return g_lz_clock; }
namespace std { namespace chrono { inline namespace _V2 {
system_clock::time_point system_clock::now() noexcept
{
    return time_point(duration(static_cast<rep>(g_lz_clock++)));
}
} } }
"""

SETUP = """\
import sys
from setuptools import setup, Extension
from Cython.Build import cythonize
import os
extra = ['-std=c++11', '-O2']
if os.environ.get('LZ_ORACLE_DET_HEADER'):
    extra += ['-include', os.environ['LZ_ORACLE_DET_HEADER']]
det = bool(os.environ.get('LZ_ORACLE_DET_HEADER'))
exts = [
    Extension('lzero.mcts.ctree.ctree_efficientzero.ez_tree',
              ['lzero/mcts/ctree/ctree_efficientzero/ez_tree.pyx'], language='c++', extra_compile_args=extra),
    Extension('lzero.mcts.ctree.ctree_muzero.mz_tree',
              ['lzero/mcts/ctree/ctree_muzero/mz_tree.pyx'], language='c++', extra_compile_args=extra),
    # Sampled EfficientZero: the deterministic flavour also links oracle_clock.cpp, which replaces
    # std::chrono::system_clock::now() (the seed of the std::default_random_engine built inside CNode::expand,
    # ctree_sampled_efficientzero/lib/cnode.cpp:251,336) by a settable counter
    Extension('lzero.mcts.ctree.ctree_sampled_efficientzero.ezs_tree',
              ['lzero/mcts/ctree/ctree_sampled_efficientzero/ezs_tree.pyx'] + (['oracle_clock.cpp'] if det else []),
              language='c++', extra_compile_args=extra, extra_link_args=(['-Wl,-Bsymbolic-functions'] if det else [])),
    # Gumbel MuZero: deterministic by construction (every node's Gumbel vector comes from std::mt19937(0), cnode.cpp:1133-1151)
    Extension('lzero.mcts.ctree.ctree_gumbel_muzero.gmz_tree',
              ['lzero/mcts/ctree/ctree_gumbel_muzero/gmz_tree.pyx'], language='c++', extra_compile_args=extra),
]
setup(ext_modules=cythonize(exts, language_level=3))
"""


def built(flavour):
    d = os.path.join(OUT, flavour)
    if not os.path.isdir(d):
        return False
    names = os.listdir(d)
    return all(any(n.startswith(stem) and n.endswith(".so") for n in names) for stem in ("ez_tree", "mz_tree", "ezs_tree", "gmz_tree"))


def build(force=False):
    if not os.path.isdir(os.path.join(REF, "lzero", "mcts", "ctree")):
        return False  # GPU box: only the prebuilt files travel
    if not force and built("stock") and built("det"):
        return True
    for flavour in ("stock", "det"):
        tmp = tempfile.mkdtemp(prefix="lz_ref_")
        try:
            dst = os.path.join(tmp, "lzero", "mcts")
            os.makedirs(dst)
            shutil.copytree(os.path.join(REF, "lzero", "mcts", "ctree"), os.path.join(dst, "ctree"))
            for root, _, files in os.walk(tmp):
                os.chmod(root, 0o755)
                for f in files:
                    os.chmod(os.path.join(root, f), 0o644)
            open(os.path.join(tmp, "lzero", "__init__.py"), "w").close()
            open(os.path.join(tmp, "lzero", "mcts", "__init__.py"), "w").close()
            with open(os.path.join(tmp, "setup_probe.py"), "w") as f:
                f.write(SETUP)
            cmd = [sys.executable, "setup_probe.py", "build_ext", "--inplace"]
            env = dict(os.environ)
            env.pop("LZ_ORACLE_DET_HEADER", None)
            if flavour == "det":
                hdr = os.path.join(tmp, "oracle_det.h")
                with open(hdr, "w") as f:
                    f.write(DET_HEADER)
                with open(os.path.join(tmp, "oracle_clock.cpp"), "w") as f:
                    f.write(CLOCK_CPP)
                env["LZ_ORACLE_DET_HEADER"] = hdr
            subprocess.run(cmd, cwd=tmp, check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            odir = os.path.join(OUT, flavour)
            os.makedirs(odir, exist_ok=True)
            for sub in ("ctree_efficientzero", "ctree_muzero", "ctree_sampled_efficientzero", "ctree_gumbel_muzero"):
                d = os.path.join(dst, "ctree", sub)
                for n in os.listdir(d):
                    if n.endswith(".so"):
                        shutil.copy(os.path.join(d, n), os.path.join(odir, n))
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return True


def load(flavour="det"):
    """Import (ez_tree, mz_tree) of the given flavour from oracle/_ref; None if not built.  ``load_sampled``
    returns the Sampled-EfficientZero module."""
    import importlib.machinery
    import importlib.util
    d = os.path.join(OUT, flavour)
    mods = []
    for stem, dotted in (("ez_tree", "lzero.mcts.ctree.ctree_efficientzero.ez_tree"),
                         ("mz_tree", "lzero.mcts.ctree.ctree_muzero.mz_tree")):
        if not os.path.isdir(d):
            return None
        cands = [n for n in os.listdir(d) if n.startswith(stem) and n.endswith(".so")]
        if not cands:
            return None
        # each flavour gets a private module name so stock and det can coexist in one process
        name = "%s_%s" % (stem, flavour)
        loader = importlib.machinery.ExtensionFileLoader(stem, os.path.join(d, cands[0]))
        spec = importlib.util.spec_from_file_location(stem, os.path.join(d, cands[0]), loader=loader)
        mod = importlib.util.module_from_spec(spec)
        loader.exec_module(mod)
        mods.append(mod)
    return tuple(mods)


def load_sampled(flavour="det"):
    """(ezs_tree module, ctypes handle exposing oracle_set_clock / oracle_get_clock for the det flavour)."""
    import ctypes
    import importlib.machinery
    import importlib.util
    d = os.path.join(OUT, flavour)
    if not os.path.isdir(d):
        return None
    cands = [n for n in os.listdir(d) if n.startswith("ezs_tree") and n.endswith(".so")]
    if not cands:
        return None
    path = os.path.join(d, cands[0])
    loader = importlib.machinery.ExtensionFileLoader("ezs_tree", path)
    spec = importlib.util.spec_from_file_location("ezs_tree", path, loader=loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    handle = None
    if flavour == "det":
        handle = ctypes.CDLL(path)
        handle.oracle_set_clock.argtypes = [ctypes.c_uint64]
        handle.oracle_get_clock.restype = ctypes.c_uint64
    return mod, handle


def load_gumbel(flavour="stock"):
    """gmz_tree module of the reference (lzero/mcts/ctree/ctree_gumbel_muzero); None if not built."""
    import importlib.machinery
    import importlib.util
    d = os.path.join(OUT, flavour)
    if not os.path.isdir(d):
        return None
    cands = [n for n in os.listdir(d) if n.startswith("gmz_tree") and n.endswith(".so")]
    if not cands:
        return None
    path = os.path.join(d, cands[0])
    loader = importlib.machinery.ExtensionFileLoader("gmz_tree", path)
    spec = importlib.util.spec_from_file_location("gmz_tree", path, loader=loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv)
    print("oracle/_ref built" if ok else "reference not present; nothing built")
