"""GPU parity (through the C ABI): the Gumbel MuZero tree kernels vs the CPU oracle and the golden vectors generated from the
reference's own compiled gmz_tree -- bit-exact records, visit counts, root values, improved policies, completed values."""
import os

import numpy as np
import pytest

import gumbel_driver as gd

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _dev_mod(c):
    from lightzero_amd.mcts.ctree.ctree_gumbel_muzero import gmz_tree as m

    def mk(n, legal, **kw):
        return m.Roots(n, legal, action_space_size=c["A"], max_simulations=c["S"])
    return type("M", (), dict(Roots=staticmethod(mk), MinMaxStatsList=m.MinMaxStatsList, ResultsWrapper=m.ResultsWrapper,
                              batch_traverse=staticmethod(m.batch_traverse), batch_back_propagate=staticmethod(m.batch_back_propagate)))


@pytest.mark.parametrize("name", sorted(gd.CASES))
def test_device_gumbel_tree_matches_oracle_and_golden(name):
    from oracle import ctree as octree
    c = gd.make_inputs(gd.CASES[name])
    dev = gd.run_tree(_dev_mod(c), c)
    ora = gd.run_tree(octree.gmz_tree, c, roots_kwargs=dict(action_space_size=c["A"], max_simulations=c["S"]))
    gd.assert_same(ora, dev, name)
    g = np.load(os.path.join(GOLD, "gumbel_%s.npz" % name))
    assert np.array_equal(dev["records"], g["records"])
    assert np.array_equal(dev["policies"].view(np.uint32), g["policies"].view(np.uint32))


def test_gumbel_fused_search_and_policy_vs_oracle_pipeline():
    """GumbelMuZeroMCTSCtree.search with the engine MuZero model (whole loop on the device) and with a torch model driving the
    device tree, vs the oracle pipeline; then the policy surface (gumbel_muzero.py:594-603 output contract)."""
    import torch
    from oracle import ctree as octree, search as osearch, torch_models as tm
    from lightzero_amd.mcts.tree_search.mcts_ctree import GumbelMuZeroMCTSCtree
    from lightzero_amd.model.muzero_model import MuZeroModel
    from lightzero_amd.policy.gumbel_muzero import GumbelMuZeroPolicy
    B, A, S = 32, 6, 30
    cfg = dict(num_simulations=S, discount_factor=0.997, max_num_considered_actions=4, value_delta_max=0.01, root_noise_weight=0.25,
               root_dirichlet_alpha=0.3)
    ref = tm.synthetic_init(tm.MuZeroModel(action_space_size=A))
    model = MuZeroModel(action_space_size=A).load_state_dict(ref.state_dict())
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(41))
    rng = np.random.default_rng(9)
    mask = (rng.random((B, A)) < 0.7).astype(np.float32); mask[:, 3] = 1
    legal = [np.nonzero(m)[0].tolist() for m in mask]
    noises = [rng.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal]
    ist = tm.InverseScalarTransform()
    with torch.no_grad():
        o = ref.initial_inference(obs)
    pred = ist(o.value).reshape(-1).numpy().tolist()
    pol = o.policy_logits.numpy().tolist()
    oroots = octree.gmz_tree.Roots(B, legal, action_space_size=A, max_simulations=S)
    oroots.prepare(0.25, noises, [0.0] * B, pred, pol, [-1] * B)
    osearch.gmz_search(octree.gmz_tree, oroots, ref, o.latent_state.numpy(), [-1] * B, cfg)
    mcts = GumbelMuZeroMCTSCtree(cfg)
    # torch model + device tree through the reference loop: the tree sees the same floats as the oracle's -> identical
    roots_b = mcts.roots(B, legal, action_space_size=A, max_simulations=S)
    roots_b.prepare(0.25, noises, [0.0] * B, pred, pol, [-1] * B)
    mcts.search(roots_b, ref, o.latent_state.numpy(), [-1] * B)
    assert roots_b.get_distributions() == oroots.get_distributions()
    assert np.array_equal(np.asarray(roots_b.get_policies(0.997, A), np.float32).view(np.uint32),
                          np.asarray(oroots.get_policies(0.997, A), np.float32).view(np.uint32))
    # engine model, fused
    roots_a = mcts.roots(B, legal, action_space_size=A, max_simulations=S)
    out = model.initial_inference(obs.cuda().contiguous(), roots_a)
    roots_a.prepare_from_inference(0.25, noises, [-1] * B)
    mcts.search(roots_a, model, out.latent_state, [-1] * B)
    same = sum(int(x == y) for x, y in zip(roots_a.get_distributions(), oroots.get_distributions()))
    import parity_record
    parity_record.record("e2e/gmz_atari96/B%d_S%d" % (B, S), {}, extra=dict(roots=B, identical_visit_distributions=same, identical_fraction=same / float(B), gate=0.85))
    assert same >= int(0.85 * B), "only %d / %d visit-count distributions identical" % (same, B)
    # policy surface
    policy = GumbelMuZeroPolicy(cfg, model)
    res = policy._forward_collect(obs.cuda().contiguous(), action_mask=mask, temperature=1.0, to_play=[-1] * B)
    r0 = res[0]
    assert set(r0) == {"action", "visit_count_distributions", "visit_count_distribution_entropy", "searched_value",
                       "roots_completed_value", "improved_policy_probs", "predicted_value", "predicted_policy_logits"}
    for i in range(B):
        assert mask[i][res[i]["action"]] == 1 and sum(res[i]["visit_count_distributions"]) == S
        assert abs(float(np.sum(res[i]["improved_policy_probs"])) - 1.0) < 1e-4
    ev = policy._forward_eval(obs.cuda().contiguous(), action_mask=mask, to_play=[-1] * B)
    ev2 = policy._forward_eval(obs.cuda().contiguous(), action_mask=mask, to_play=[-1] * B)
    assert all(ev[i]["action"] == ev2[i]["action"] for i in range(B))
