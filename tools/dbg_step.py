"""Debugging aid: steps the device tree (fine-grained API) and the C oracle side by side on a recorded case and reports the
first simulation after which any observable differs (records, min-max stats, root value sums, visit distributions)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import ctree as octree
from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree
g = np.load(os.path.join(ROOT, "tools", "_dbg_case.npz"))
S, B, A = g["logits"].shape
sel = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else list(range(B))
B2 = len(sel)
legal = [list(range(A))] * B2
def mk(mod, **kw):
    r = mod.Roots(B2, legal, **kw)
    r.prepare(0.25, g["noises"][sel].tolist(), [0.0] * B2, g["root_logits"][sel].tolist(), [-1] * B2)
    mm = mod.MinMaxStatsList(B2); mm.set_delta(0.01)
    return r, mm
dr, dmm = mk(mz_tree, action_space_size=A, max_simulations=S); dr.set_tiebreak(0)
orr, omm = mk(octree.mz_tree, action_space_size=A, max_simulations=S)
for s in range(S):
    rd, ro = mz_tree.ResultsWrapper(B2), octree.mz_tree.ResultsWrapper(B2)
    a = mz_tree.batch_traverse(dr, 19652, 1.25, 0.997, dmm, rd, [-1] * B2, True)
    b = octree.mz_tree.batch_traverse(orr, 19652, 1.25, 0.997, omm, ro, [-1] * B2)
    ra = np.array([a[0], a[2], rd.get_search_len()]); rb = np.array([b[0], b[2], ro.get_search_len()])
    if not np.array_equal(ra, rb):
        bad = np.nonzero((ra != rb).any(0))[0]
        print("sim", s, "traverse differs on roots", [sel[i] for i in bad[:8]], "device", ra[:, bad[0]].tolist(), "oracle", rb[:, bad[0]].tolist())
        td_, to_ = dr.get_trajectories()[bad[0]], orr.get_trajectories()[bad[0]]
        k = next((i for i, (x, y) in enumerate(zip(td_, to_)) if x != y), min(len(td_), len(to_)))
        print("trajectory lengths", len(td_), len(to_), "first differing depth", k, "device", td_[max(0, k - 3):k + 3], "oracle", to_[max(0, k - 3):k + 3])
        break
    vp, v, lg = g["vp"][s][sel], g["v"][s][sel], g["logits"][s][sel]
    mz_tree.batch_backpropagate(s + 1, 0.997, vp.tolist(), v.tolist(), lg.tolist(), dmm, rd, a[3])
    octree.mz_tree.batch_backpropagate(s + 1, 0.997, vp.tolist(), v.tolist(), lg.tolist(), omm, ro, b[3])
    m1, m2 = dr.get_minmax(), orr.get_minmax()
    v1, v2 = np.asarray(dr.get_values(), np.float32), np.asarray(orr.get_values(), np.float32)
    if not np.array_equal(m1.view(np.uint32), m2.view(np.uint32)) or not np.array_equal(v1.view(np.uint32), v2.view(np.uint32)):
        bad = np.nonzero((m1 != m2).any(1) | (v1 != v2))[0]
        i = bad[0]
        print("sim", s, "state differs after backprop on roots", [sel[k] for k in bad[:8]], "minmax dev", m1[i].tolist(), "oracle", m2[i].tolist(),
              "root value dev %r oracle %r" % (float(v1[i]), float(v2[i])), "search_len", rd.get_search_len()[i], "leaf vp/v", float(vp[i]), float(v[i]))
        break
else:
    print("no difference in", S, "simulations")
