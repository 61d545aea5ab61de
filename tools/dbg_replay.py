import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import tree_driver as td
import test_exact_replay_gpu as T
from lightzero_amd import _lib as L
from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree
from oracle import ctree as octree
lib = L.lib()
B, A = int(sys.argv[1]), 4
S = int(sys.argv[2])
model = T._mz_model(A, seed=3)
obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(23)).cuda().contiguous()
rng = np.random.default_rng(5)
noises = [rng.dirichlet([0.3] * A).astype(np.float32).tolist() for _ in range(B)]
legal = [list(range(A))] * B
roots = mz_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
roots.set_tiebreak(0)
model.initial_inference(obs, roots, fetch=False)
L.check(lib.lz_roots_enable_trace(roots._h, 1))
roots.prepare_from_inference(0.25, noises, [-1] * B)
L.check(lib.lz_search(roots._h, S, 19652, 1.25, 0.997, 0, 0.01))
dist, cnt, val, pred, logits0 = roots.get_search_results()
sims = []
for s in range(1, S + 1):
    vp = np.zeros(B, np.float32); v = np.zeros(B, np.float32); lg = np.zeros((B, A), np.float32)
    L.check(lib.lz_roots_read_sim_outputs(roots._h, s, vp, v, lg.reshape(-1)))
    sims.append(dict(vp=vp, v=v, logits=lg))
tr = np.zeros((S, B, 4), np.int32)
L.check(lib.lz_roots_read_trace(roots._h, S, tr.reshape(-1)))
case = dict(variant="mz", B=B, A=A, S=S, legal_list=legal, to_play_list=[-1] * B, root_logits=logits0, root_vp=np.zeros(B, np.float32),
            noises=noises, noise_w=0.25, sims=sims, discount=0.997, **T.PB)
o = td.run_tree(octree.mz_tree, case, roots_kwargs=dict(action_space_size=A, max_simulations=S))
rec = o["records"][:, :, [0, 2, 3, 4]]
bad = np.argwhere((rec != tr).any(2))
print("env", os.environ.get("LZ_NO_TREE_FUSE"), os.environ.get("LZ_TREE_NO_LDS"), "B", B, "S", S, "mismatching (sim, root) pairs:", len(bad))
if len(bad):
    s0 = bad[:, 0].min()
    rows = bad[bad[:, 0] == s0][:5]
    print("first divergent simulation", s0, "roots", rows[:, 1].tolist())
    for s, b in rows:
        print("  oracle", rec[s, b].tolist(), "device", tr[s, b].tolist(), "prev sim outputs vp/v", sims[s - 1]["vp"][b], sims[s - 1]["v"][b])
    print("finite:", all(np.isfinite(x["v"]).all() and np.isfinite(x["vp"]).all() and np.isfinite(x["logits"]).all() for x in sims))
from oracle import build_ref
ref = build_ref.load("det")
if ref:
    o2 = td.run_tree(ref[1], case)
    print("C oracle vs compiled reference: records equal", np.array_equal(o2["records"], o["records"]), "dists equal", o2["distributions"] == o["distributions"])
dev_tree = td.run_tree(mz_tree, case, roots_kwargs=dict(action_space_size=A, max_simulations=S, engine=model.engine), traverse_kwargs=dict(deterministic=True))
print("device tree-only (separate traverse / backprop kernels) vs C oracle: records equal", np.array_equal(dev_tree["records"], o["records"]),
      "; vs the fused search's trace:", np.array_equal(dev_tree["records"][:, :, [0, 2, 3, 4]], tr))
out = os.path.join(ROOT, "gpurun_out", "dbg_case_B%d_S%d.npz" % (B, S))
os.makedirs(os.path.dirname(out), exist_ok=True)
if B <= 64:
    np.savez_compressed(out, root_logits=logits0, noises=np.asarray(noises, np.float32), vp=np.stack([x["vp"] for x in sims]), v=np.stack([x["v"] for x in sims]),
                        logits=np.stack([x["logits"] for x in sims]), trace=tr, dev_dist=dist, dev_val=val, dev_tree_rec=dev_tree["records"])
same = sum(int(a == b) for a, b in zip(o["distributions"], [dist[i, :cnt[i]].tolist() for i in range(B)]))
print("identical dists", same, "/", B, "max search len", tr[:, :, 2].max())
