"""GPU: two host threads, each with its own model (= its own engine / HIP stream) and roots, search concurrently -- the results must be
the ones each thread gets alone (deterministic tie-break, no noise): per-engine state is not shared, the library's global state (error
text, parked-handle cache) is not corrupted by concurrent callers."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _work(model, obs, legal, S, reps, out):
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    try:
        torch.cuda.set_device(0)
        res = []
        for _ in range(reps):
            B, A = obs.shape[0], model.action_space_size
            roots = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
            roots.set_tiebreak(0)
            model.initial_inference(obs, roots, fetch=False)
            roots.prepare_from_inference_no_noise([-1] * B)
            L.check(L.lib().lz_search(roots._h, S, 19652, 1.25, 0.997, 5, 0.01))
            res.append((roots.get_distributions(), np.asarray(roots.get_values(), np.float32).tobytes()))
        out.append(res)
    except Exception as e:   # surfaced by the main thread
        out.append(e)


def test_two_threads_two_engines_search_concurrently():
    from oracle import torch_models as tm
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    jobs = []
    for k, (A, B, S) in enumerate(((6, 96, 30), (9, 57, 22))):
        sd = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A), seed=10 + k).state_dict()
        model = EfficientZeroModel(action_space_size=A).load_state_dict(sd)
        obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(k)).cuda().contiguous()
        jobs.append((model, obs, [list(range(A))] * B, S))
    alone = []
    for j in jobs:
        out = []
        _work(*j, 2, out)
        assert not isinstance(out[0], Exception), out[0]
        alone.append(out[0])
        assert alone[-1][0] == alone[-1][1]   # deterministic from run to run
    outs = [[], []]
    th = [threading.Thread(target=_work, args=(*jobs[i], 6, outs[i])) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
        assert not t.is_alive()
    for i in range(2):
        assert not isinstance(outs[i][0], Exception), outs[i][0]
        assert all(r == alone[i][0] for r in outs[i][0]), "thread %d: results differ from the single-threaded run" % i
