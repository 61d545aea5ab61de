#!/usr/bin/env python
"""Debugging aid (debug build of the library): cycle stamps of the SEPARATE tree step k_backprop_traverse (the HBM walk of deep trees,
BASELINE configs[2]) for 64 roots of the last launch of a search -- table fill, expand + backup, selection -- with the path depths.
    python tools/tree_sep_timing.py [sims]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LZ_DEBUG_TREE_SEP_TS", "1"); os.environ.setdefault("LZ_NO_GRAPH", "1")
from lightzero_amd import build as _b
os.environ.setdefault("LZ_MI355_LIB", _b.DBG_LIB)
import torch
from lightzero_amd import _lib as L
from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree
from lightzero_amd.model.muzero_model import MuZeroModel
from lightzero_amd.model.synthetic import efficientzero_state_dict
lib = L.lib()
A, B, S = 4, 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 400
model = MuZeroModel(action_space_size=A).load_state_dict(efficientzero_state_dict(seed=0, action_space_size=A, muzero=True))
roots = mz_tree.Roots(B, [list(range(A))] * B, action_space_size=A, max_simulations=S, engine=model.engine); roots._ensure(A)
obs = torch.rand(B, 4, 96, 96).cuda()
for it in range(2):
    L.check(lib.lz_initial_inference(roots._h, obs.data_ptr()))
    L.check(lib.lz_roots_prepare_from_inference(roots._h, 0.25, None, L.i32([-1] * B)))
    L.check(lib.lz_search(roots._h, S, 19652, 1.25, 0.997, 0, 0.01))
    L.check(lib.lz_engine_synchronize(model.engine))
lib.lz_debug_read_tree_sep_ts.argtypes = [ctypes.c_void_p]
out = np.zeros((64, 8), np.uint64)
L.check(lib.lz_debug_read_tree_sep_ts(out.ctypes.data))
t = out.astype(np.int64)
fill, back, sel = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
dprev, dnext = t[:, 4], t[:, 5]
print("64 roots of the last k_backprop_traverse launch (simulation %d): cycles" % (S - 1))
print("  table fill   mean %7.0f" % fill.mean())
print("  expand+backup mean %7.0f  per level of the backed-up path %6.0f  (path depth mean %.1f, max %d)" % (back.mean(), (back / np.maximum(dprev, 1)).mean(), dprev.mean(), dprev.max()))
print("  selection    mean %7.0f  per level %6.0f  (depth mean %.1f, max %d)" % (sel.mean(), (sel / np.maximum(dnext, 1)).mean(), dnext.mean(), dnext.max()))
print("  total        mean %7.0f  max %7.0f" % ((t[:, 3] - t[:, 0]).mean(), (t[:, 3] - t[:, 0]).max()))
