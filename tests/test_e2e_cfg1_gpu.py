"""GPU, BASELINE configs[1] at full size (256 roots x 50 simulations, EfficientZero Atari 96x96x4): the fused device search against
the FULL oracle pipeline (reference-style driver + torch fp32 model + CPU ctree) on the same observations, weights and Dirichlet noise,
deterministic tie-break on both sides.

Recorded (profiles/rNN_parity.json, entry "e2e/ez_atari96/B256_S50") and gated at >= 98 % (sharp-prior case: 95 %): the fraction of roots whose visit
distributions coincide, and for EVERY differing root the first simulation whose selection differed with the pUCT scores of the two
competing actions on both sides (tests/e2e_attrib.py).  The two pipelines feed their trees network outputs that differ by <= 1e-5
before the inverse scalar transform (tests/test_nn_gpu.py) and by whole steps of the reference formula's own ~1.3e-4 quantum after it;
the tree itself is bit-exact on identical inputs (tests/test_exact_replay_gpu.py).  So a root may differ only where two scores were
that close: every differing root is classified, and an "unexplained" one fails the test.
mcts_ctree.py:839-842 (the scalars the tree is fed), cnode.cpp:651-695 / 756-814 (the selection)."""
import json
import os

import numpy as np
import pytest
import torch

import e2e_attrib
import parity_record

pytestmark = pytest.mark.gpu

CFG = dict(num_simulations=50, pb_c_base=19652, pb_c_init=1.25, discount_factor=0.997, value_delta_max=0.01,
           lstm_horizon_len=5, root_noise_weight=0.25, root_dirichlet_alpha=0.3)


def _device_records(roots, lib, L, B, A, S):
    trace = np.zeros((S, B, 4), np.int32)
    L.check(lib.lz_roots_read_trace(roots._h, S, trace.reshape(-1)))
    rec = []
    for s in range(S):
        vp = np.zeros(B, np.float32); val = np.zeros(B, np.float32); pol = np.zeros((B, A), np.float32)
        L.check(lib.lz_roots_read_sim_outputs(roots._h, s + 1, vp, val, pol.reshape(-1)))
        rec.append(dict(ix=trace[s, :, 0].copy(), action=trace[s, :, 1].copy(), search_len=trace[s, :, 2].copy(), value_prefix=vp, value=val, policy_logits=pol))
    return rec


@pytest.mark.parametrize("B,S,seed,sharp", [(256, 50, 5, 1.0), (256, 50, 11, 1.0), (256, 50, 7, 10.0)])
def test_cfg1_full_size_vs_oracle_pipeline(B, S, seed, sharp):
    """sharp = 10: the policy head's and the value head's last layers x 10 (lightzero_amd.model.synthetic.sharpen_state_dict) -- a
    trained agent's sharp prior: root max-prob ~0.9 instead of ~0.24, search paths of depth 4-10 instead of 2-3 (VERDICT r4 weak #3:
    the 8d recipe only ever builds the shallowest trees)."""
    from oracle import ctree as octree, search as osearch, torch_models as tm
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    from lightzero_amd.model.synthetic import sharpen_state_dict
    A = 6
    cfg = dict(CFG, num_simulations=S)
    ref = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A))
    if sharp != 1.0:
        ref.load_state_dict(sharpen_state_dict(ref.state_dict(), sharp))
    model = EfficientZeroModel(action_space_size=A).load_state_dict(ref.state_dict())
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(seed))
    rng = np.random.default_rng(seed)
    noises = [rng.dirichlet([0.3] * A).astype(np.float32).tolist() for _ in range(B)]
    legal = [list(range(A))] * B
    rec_o = []
    o_dist, o_val, o_pred, o_logits = osearch.ez_forward_collect(
        octree.ez_tree, ref, obs, legal, noises, [-1] * B, cfg, roots_kwargs=dict(action_space_size=A, max_simulations=S), record=rec_o)
    lib = L.lib()
    roots = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S)
    roots.set_tiebreak(0)
    out = model.initial_inference(obs.cuda().contiguous(), roots)
    roots.prepare_from_inference(cfg["root_noise_weight"], noises, [-1] * B)
    L.check(lib.lz_roots_enable_trace(roots._h, 1))
    L.check(lib.lz_search(roots._h, S, cfg["pb_c_base"], cfg["pb_c_init"], cfg["discount_factor"], cfg["lstm_horizon_len"], cfg["value_delta_max"]))
    d_dist, d_val = roots.get_distributions(), np.array(roots.get_values())
    rec_d = _device_records(roots, lib, L, B, A, S)
    d_logits = np.asarray(out.policy_logits, np.float32)
    same = np.array([a == b for a, b in zip(o_dist, d_dist)])
    entries, benign = [], 0
    for b in range(B):
        e = e2e_attrib.attribute(octree.ez_tree, cfg, A, legal[b], noises[b], o_logits[b], d_logits[b], rec_o, rec_d, b)
        if e is None:
            assert same[b]
            continue
        if same[b]:
            benign += 1   # a selection differed somewhere and the visit counts still coincide
        else:
            entries.append(e)
    summ = e2e_attrib.summarize(entries, B, int(same.sum()))
    summ["roots_with_a_differing_selection_but_identical_visit_counts"] = benign
    summ["root_value_max_abs_diff_on_identical_roots"] = float(np.abs(np.array(o_val) - d_val)[same].max())
    summ["differing_roots"] = entries
    print(json.dumps({k: v for k, v in summ.items() if k != "differing_roots"}))
    for e in entries:
        print("  root %(root)d: first differing simulation %(first_sim)d, level %(level)d, gaps %(gap_oracle).3g / %(gap_device).3g, "
              "best two (oracle) %(best_two_oracle).3g, scalar delta %(scalar_delta).3g, range %(minmax_range).3g -> %(class)s" % e)
    depth = np.stack([r["search_len"] for r in rec_d])   # [S][B]
    summ["search_depth_mean"], summ["search_depth_max"] = float(depth.mean()), int(depth.max())
    parity_record.record("e2e/ez_atari96/B%d_S%d/seed%d%s" % (B, S, seed, "" if sharp == 1.0 else "/sharp%g" % sharp), {}, extra=summ)
    # Gated at the evidence: 256 / 256 identical roots were measured on two seeds with the 8d recipe, so the gate is
    # 98 % there; the sharp-prior case walks deeper (more recurrent steps between the two pipelines' scalars) and is held to 95 %.
    # A differing root that is NOT a near-tie or one quantum of the post-transform scalars fails the test.
    assert same.mean() >= (0.98 if sharp == 1.0 else 0.95), "only %d / %d roots have identical visit distributions" % (int(same.sum()), B)
    if sharp != 1.0:
        assert depth.max() >= 5, "the sharp-prior case did not build deeper trees (max search length %d)" % depth.max()
    bad = [e for e in entries if e["class"] == "unexplained"]
    assert not bad, "differing roots not explained by a near-tie or one quantum of the post-transform scalars: %r" % bad
