"""CPU: the attribution used by tests/test_e2e_cfg1_gpu.py (tests/e2e_attrib.py) on two CPU pipelines -- the oracle pipeline and the same
pipeline with every weight perturbed at the 2e-4 level (a hundred times what a different summation order does to the network outputs, so that roots do flip): for every
root whose visit distributions differ the first differing selection is found, each side's replayed tree reproduces its own records,
the probes end in the recorded selections, and the competing scores are close on both sides."""
import copy

import numpy as np
import torch

import e2e_attrib

CFG = dict(num_simulations=40, pb_c_base=19652, pb_c_init=1.25, discount_factor=0.997, value_delta_max=0.01,
           lstm_horizon_len=5, root_noise_weight=0.25, root_dirichlet_alpha=0.3)


def test_attribution_of_flipped_roots():
    from oracle import ctree as octree, search as osearch, torch_models as tm
    B, A, S = 48, 6, CFG["num_simulations"]
    ref = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A))
    other = copy.deepcopy(ref)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for p in other.parameters():
            p.mul_(1.0 + 2e-4 * torch.randn(p.shape, generator=g))
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(5))
    rng = np.random.default_rng(0)
    noises = [rng.dirichlet([0.3] * A).astype(np.float32).tolist() for _ in range(B)]
    legal = [list(range(A))] * B
    kw = dict(roots_kwargs=dict(action_space_size=A, max_simulations=S))
    rec_a, rec_b = [], []
    d_a, _, _, lg_a = osearch.ez_forward_collect(octree.ez_tree, ref, obs, legal, noises, [-1] * B, CFG, record=rec_a, **kw)
    d_b, _, _, lg_b = osearch.ez_forward_collect(octree.ez_tree, other, obs, legal, noises, [-1] * B, CFG, record=rec_b, **kw)
    same = [x == y for x, y in zip(d_a, d_b)]
    entries = []
    for b in range(B):
        e = e2e_attrib.attribute(octree.ez_tree, CFG, A, legal[b], noises[b], lg_a[b], lg_b[b], rec_a, rec_b, b)
        if not same[b]:
            assert e is not None   # different visit counts need a different selection somewhere
        if e is not None:
            assert e["gap_oracle"] >= 0.0 and e["gap_device"] >= 0.0 and 0 <= e["first_sim"] < S
            entries.append(e)
    summ = e2e_attrib.summarize([e for e in entries if not same[e["root"]]], B, sum(same))
    print(summ, [(e["root"], e["first_sim"], e["level"], e["gap_oracle"], e["gap_device"], e["class"]) for e in entries])
    assert entries, "the perturbation flipped no selection: the test exercises nothing"
    assert summ["differing"] == B - sum(same)
