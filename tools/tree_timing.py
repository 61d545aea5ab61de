#!/usr/bin/env python
"""Debugging aid: s_memtime stamps of root 0's tree step inside the last tree-fused chain launch of a search (debug build of the library).
    python tools/tree_timing.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LZ_DEBUG_TREE_TS", "1"); os.environ.setdefault("LZ_NO_GRAPH", "1")   # (the stamp buffer is allocated at the first launch: not inside a capture)
from lightzero_amd import build as _b
os.environ.setdefault("LZ_MI355_LIB", _b.DBG_LIB)
assert os.path.exists(os.environ["LZ_MI355_LIB"]), "build the debug library first: python -m lightzero_amd.build --debug-knobs"
import torch
from lightzero_amd import _lib as L
from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
from lightzero_amd.model.efficientzero_model import EfficientZeroModel
from lightzero_amd.model.synthetic import efficientzero_state_dict
lib = L.lib()
model = EfficientZeroModel(action_space_size=6, fast_mode=bool(os.environ.get("LZ_TOOL_FAST"))).load_state_dict(efficientzero_state_dict(seed=0, action_space_size=6))
B, S = 256, 50
roots = ez_tree.Roots(B, [list(range(6))] * B, action_space_size=6, max_simulations=S, engine=model.engine); roots._ensure(6)
obs = torch.rand(B, 4, 96, 96).cuda()
for it in range(3):
    L.check(lib.lz_initial_inference(roots._h, obs.data_ptr()))
    L.check(lib.lz_roots_prepare_from_inference(roots._h, 0.25, None, L.i32([-1] * B)))
    L.check(lib.lz_search(roots._h, S, 19652, 1.25, 0.997, 5, 0.01))
    L.check(lib.lz_engine_synchronize(model.engine))
lib.lz_debug_read_tree_ts.argtypes = [ctypes.c_void_p]
out = np.zeros(32, np.uint64)
L.check(lib.lz_debug_read_tree_ts(out.ctypes.data))
ts = out[:7].astype(np.int64)
names = ["step entered", "all loads requested", "tree staged in LDS", "expand + backup", "selection", "barrier (other waves released)", "latent / tables staged"]
for i in range(1, 7):
    print("%-32s +%6d cycles   (t=%6d)" % (names[i], ts[i] - ts[i - 1], ts[i] - ts[0]))
hs = out[8:15].astype(np.int64)
if hs[0]:   # split heads: head wave 1 (value head, output group 0) of the same workgroup
    hn = ["entered", "-", "all requests issued", "partials arrived + hidden units", "second layer", "own softmax sums", "rendezvous of the three waves", "scalar out"]
    print("head wave 1 (value head): entered at t=%6d of the tree wave's clock" % (hs[0] - ts[0]))
    for i in range(1, 7):
        print("  %-30s +%6d cycles   (t=%6d)" % (hn[i + 1] if i < 6 else hn[7], hs[i] - hs[i - 1], hs[i] - ts[0]))
if out[24]:   # k_chain_b (LZ_TOOL_FAST=1): ends of the layers, end of the kernel (tree wave's clock)
    prev = ts[6]
    for i in range(8):
        if out[16 + i]:
            print("layer %d done                     +%6d cycles   (t=%6d)" % (i, int(out[16 + i]) - prev, int(out[16 + i]) - ts[0]))
            prev = int(out[16 + i])
    print("head convolutions / kernel end   +%6d cycles   (t=%6d)" % (int(out[24]) - prev, int(out[24]) - ts[0]))
if out[25]:
    print("layer 2: weights requested t=%d, MFMAs issued t=%d, barrier 1 passed t=%d, layer end t=%d" % tuple(int(out[i]) - int(ts[0]) for i in (25, 26, 27, 18)))
if out[28]:
    print("layer 2 epilogue: reads done t=%d, writes issued t=%d, next weights in registers t=%d" % tuple(int(out[i]) - int(ts[0]) for i in (28, 29, 30)))
elif out[29]:
    print("layer 2 epilogue: writes issued t=%d" % (int(out[29]) - int(ts[0])))
