#!/usr/bin/env python
"""Debugging aid: per-phase time stamps (s_memtime = shader cycles on gfx950, ~2.0 GHz under this load) of one k_chain workgroup during a search.
    LZ_DEBUG_CHAIN_TS=1 LZ_NO_GRAPH=1 python tools/chain_timing.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LZ_DEBUG_CHAIN_TS", "1"); os.environ.setdefault("LZ_NO_GRAPH", "1")
# the stamps exist only in the -DLZ_DEBUG_KNOBS build of the library (python -m lightzero_amd.build --debug-knobs)
from lightzero_amd import build as _b
os.environ.setdefault("LZ_MI355_LIB", _b.DBG_LIB)
assert os.path.exists(os.environ["LZ_MI355_LIB"]), "build the debug library first: python -m lightzero_amd.build --debug-knobs"
import torch
from lightzero_amd import _lib as L
from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
from lightzero_amd.model.efficientzero_model import EfficientZeroModel
lib = L.lib()
from lightzero_amd.model.synthetic import efficientzero_state_dict
model = EfficientZeroModel(action_space_size=6).load_state_dict(efficientzero_state_dict(seed=0, action_space_size=6))
B, S = 256, 50
roots = ez_tree.Roots(B, [list(range(6))] * B, action_space_size=6, max_simulations=S, engine=model.engine); roots._ensure(6)
obs = torch.rand(B, 4, 96, 96).cuda()
for it in range(3):
    L.check(lib.lz_initial_inference(roots._h, obs.data_ptr()))
    L.check(lib.lz_roots_prepare_from_inference(roots._h, 0.25, None, L.i32([-1] * B)))
    L.check(lib.lz_search(roots._h, S, 19652, 1.25, 0.997, 5, 0.01))
    L.check(lib.lz_engine_synchronize(model.engine))
lib.lz_debug_read_chain_ts.argtypes = [ctypes.c_void_p]
out = np.zeros(64, np.uint64)
L.check(lib.lz_debug_read_chain_ts(out.ctypes.data))
n = int(out[0]); ts = out[1:1 + n].astype(np.int64)
if n > 30:   # Winograd chain (k_chain_w), timing instance
    names = ["staged", "sync"] + [x for l in range(5) for x in ("L%d patch read" % l, "L%d V written" % l, "L%d epi operands" % l, "L%d barrier" % l, "L%d products" % l, "L%d row sums" % l, "L%d barrier " % l, "L%d combine" % l, "L%d barrier  " % l)] + ["end"]
else:
    names = ["staged", "sync"] + [x for l in range(5) for x in ("L%d loop start" % l, "L%d loop end" % l, "L%d epilogue" % l, "L%d barrier" % l)] + ["end"]
print("stamps:", n)
for i in range(1, n):
    print("%-16s +%7d cycles   (t=%7d, ~%5.1f us at 2.0 GHz)" % (names[i] if i < len(names) else i, ts[i] - ts[i - 1], ts[i] - ts[0], (ts[i] - ts[0]) / 2000.0))
