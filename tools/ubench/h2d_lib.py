"""the library's host-observation entry (lz_initial_inference_host) against the device entry: wall time of one call + engine synchronisation"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lightzero_amd import _lib as L
from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
from lightzero_amd.model.efficientzero_model import EfficientZeroModel
from lightzero_amd.model.synthetic import efficientzero_state_dict
lib = L.lib()
model = EfficientZeroModel(action_space_size=6).load_state_dict(efficientzero_state_dict(seed=0, action_space_size=6))
B = 256
roots = ez_tree.Roots(B, [list(range(6))] * B, action_space_size=6, max_simulations=50, engine=model.engine); roots._ensure(6)
obs = torch.rand(B, 4, 96, 96)
pin = obs.clone().pin_memory()
dev = obs.cuda()
def t(f, n=10):
    for _ in range(3): f()
    L.check(lib.lz_engine_synchronize(model.engine)); t0 = time.perf_counter()
    for _ in range(n): f()
    L.check(lib.lz_engine_synchronize(model.engine)); return (time.perf_counter() - t0) / n * 1e3
print("device obs      %.3f ms" % t(lambda: L.check(lib.lz_initial_inference(roots._h, dev.data_ptr()))))
print("pinned host obs %.3f ms" % t(lambda: L.check(lib.lz_initial_inference_host(roots._h, pin.numpy()))))
print("pageable host   %.3f ms" % t(lambda: L.check(lib.lz_initial_inference_host(roots._h, obs.numpy()))))
# the whole env-step with each entry: where does the host-observation step spend its time?
to_play = L.i32([-1] * B)
def full(host, src):
    if host: L.check(lib.lz_initial_inference_host(roots._h, src))
    else: L.check(lib.lz_initial_inference(roots._h, src))
    L.check(lib.lz_roots_prepare_from_inference_dirichlet(roots._h, 0.25, 0.3, to_play))
    L.check(lib.lz_search(roots._h, 50, 19652, 1.25, 0.997, 5, 0.01))
print("step, device obs      %.3f ms" % t(lambda: full(False, dev.data_ptr())))
print("step, pinned host obs %.3f ms" % t(lambda: full(True, pin.numpy())))
print("step, pageable host   %.3f ms" % t(lambda: full(True, obs.numpy())))
def stamped(src):
    L.check(lib.lz_engine_synchronize(model.engine)); t0 = time.perf_counter()
    L.check(lib.lz_initial_inference_host(roots._h, src)); t1 = time.perf_counter()
    L.check(lib.lz_engine_synchronize(model.engine)); t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3
for name, src in (("pinned", pin.numpy()), ("pageable", obs.numpy())):
    for _ in range(3): full(True, src)
    a = [stamped(src) for _ in range(5)]
    print(name, "call returns after %.3f ms, device done %.3f ms later" % (np.median([x[0] for x in a]), np.median([x[1] for x in a])))
# after a search was enqueued (the call lands behind 2.4 ms of queued work)
for name, src in (("pinned", pin.numpy()), ("pageable", obs.numpy())):
    full(True, src); L.check(lib.lz_engine_synchronize(model.engine))
    full(True, src); t0 = time.perf_counter()
    L.check(lib.lz_initial_inference_host(roots._h, src)); t1 = time.perf_counter()
    L.check(lib.lz_engine_synchronize(model.engine)); t2 = time.perf_counter()
    print(name, "behind a queued search: call returns after %.3f ms, all done %.3f ms later" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
