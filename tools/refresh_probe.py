#!/usr/bin/env python
"""What one weight refresh costs inside a stepping loop (round 5): per-step wall times of the headline step with a device-side refresh
every K steps, the host time of the load_state_dict call itself, and the device time of the refresh alone."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightzero_amd import _lib as L, shard
from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
from lightzero_amd.model.efficientzero_model import EfficientZeroModel
from lightzero_amd.model.synthetic import efficientzero_state_dict
lib = L.lib()
B, A, S = 256, 6, 50
w = efficientzero_state_dict(seed=0, action_space_size=A)
model = EfficientZeroModel(action_space_size=A).load_state_dict(w)
eng = model.engine
roots = ez_tree.Roots(B, [list(range(A))] * B, action_space_size=A, max_simulations=S, engine=eng); roots._ensure(A)
obs = torch.rand(B, 4, 96, 96).cuda()
flat = shard.flat_state_dict(w, "cuda")
tp = L.i32([-1] * B)
rows = torch.zeros(B, shard.row_width(A, 96 * 96), device="cuda")
ts = np.zeros(B, np.int32)

def step():
    L.check(lib.lz_initial_inference(roots._h, obs.data_ptr()))
    L.check(lib.lz_roots_prepare_from_inference_dirichlet(roots._h, 0.25, 0.3, tp))
    L.check(lib.lz_search(roots._h, S, 19652, 1.25, 0.997, 5, 0.01))
    h = np.zeros((B, 8 + 2 * A), np.float32); lg = np.zeros((B, A), np.float32)
    L.check(lib.lz_roots_collect_rows(roots._h, 1.0, 0, 7, None, 96 * 96, ts.ctypes.data, rows.data_ptr(), rows.shape[1], h, lg.ctypes.data))

for _ in range(5):
    step()
model.load_state_dict(flat); L.check(lib.lz_engine_synchronize(eng))
host, dev = [], []
for _ in range(10):
    L.check(lib.lz_engine_synchronize(eng))
    t0 = time.perf_counter(); model.load_state_dict(flat); t1 = time.perf_counter(); L.check(lib.lz_engine_synchronize(eng)); t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); dev.append((t2 - t0) * 1e3)
print("refresh alone: host call %.3f ms (median), call + device %.3f ms" % (np.median(host), np.median(dev)))
for K in (0, 1, 4):
    times = []
    for i in range(24):
        t0 = time.perf_counter()
        step()
        if K and (i + 1) % K == 0:
            model.load_state_dict(flat)
        times.append((time.perf_counter() - t0) * 1e3)
    print("refresh every %d: per-step ms %s  mean %.3f" % (K, " ".join("%.2f" % t for t in times[4:16]), np.mean(times[4:])))
