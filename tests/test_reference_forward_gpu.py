"""GPU: the reference's OWN collect / eval forward bodies (tests/reference_forward_collect.py: lzero/policy/efficientzero.py:539-657,
670-747, statements unchanged, imports pointed at lightzero_amd) run on an engine model -- the call order of the reference
(model.initial_inference(obs) BEFORE the roots exist; MCTSCtree.roots(n, legal_actions); roots.prepare(host lists); search(roots, model,
latent_state_roots, reward_hidden_state_roots, to_play)) with the tensors that matter staying in HBM: the search adopts the model's
inference into the prepared roots and runs the fused device loop.  Checked against the engine-native path on the same noise
(identical visit counts, bit-equal root values) and against the torch restatement of the network (tests/parity_record.py bounds)."""
import numpy as np
import pytest
import torch

import parity_record
import reference_forward_collect as rfc

pytestmark = pytest.mark.gpu


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def _policy(model, S, tiebreak_first=True, **extra):
    from lightzero_amd.mcts.tree_search.mcts_ctree import EfficientZeroMCTSCtree

    class MCTS(EfficientZeroMCTSCtree):   # the reference's classmethod signature: roots(active_collect_env_num, legal_actions)
        @classmethod
        def roots(cls, n, legal):
            r = EfficientZeroMCTSCtree.roots(n, legal)
            if tiebreak_first:
                r.set_tiebreak(0)
            return r
    rfc.MCTSCtree = MCTS
    cfg = _Cfg(num_simulations=S, discount_factor=0.997, lstm_horizon_len=5, pb_c_base=19652, pb_c_init=1.25, value_delta_max=0.01,
               root_dirichlet_alpha=0.3, root_noise_weight=0.25, mcts_ctree=True, collect_with_pure_policy=False, device="cpu",
               eps=_Cfg(eps_greedy_exploration_in_collect=False), env_type="not_board_games",
               model=_Cfg(value_support_range=(-300., 301., 1.), reward_support_range=(-300., 301., 1.), categorical_distribution=True), **extra)
    p = rfc.ReferenceForwardBodies()
    p._cfg, p._collect_model, p._eval_model = cfg, model, model
    p._mcts_collect, p._mcts_eval = MCTS(cfg), MCTS(cfg)                                    # _init_collect / _init_eval (efficientzero.py:505-535)
    p.value_inverse_scalar_transform_handle = p._mcts_collect.value_inverse_scalar_transform_handle
    return p


def test_reference_forward_bodies_run_on_the_engine_model():
    from oracle import torch_models as tm
    from lightzero_amd import _lib as L
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    B, A, S = 64, 6, 50
    ref = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A), seed=71)
    model = EfficientZeroModel(action_space_size=A).load_state_dict(ref.state_dict())
    pol = _policy(model, S)
    obs_h = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(72))
    obs = obs_h.cuda().contiguous()
    rng = np.random.default_rng(73)
    mask = (rng.random((B, A)) < 0.8).astype(np.float32)
    mask[np.arange(B), rng.integers(0, A, size=B)] = 1.0
    to_play = [-1] * B
    for step in range(3):   # the second and third forward re-arm parked device handles instead of allocating
        np.random.seed(500 + step)
        out = pol._forward_collect(obs, action_mask=mask, temperature=1.0, to_play=to_play, epsilon=0.0)
        # the engine-native path on the same noise
        np.random.seed(500 + step)
        noises = [np.random.dirichlet([0.3] * int(sum(mask[j]))).astype(np.float32).tolist() for j in range(B)]
        legal = [np.nonzero(mask[j])[0].tolist() for j in range(B)]
        roots = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
        roots.set_tiebreak(0)
        native = model.initial_inference(obs, roots)
        roots.prepare_from_inference(0.25, noises, to_play)
        L.check(L.lib().lz_search(roots._h, S, 19652, 1.25, 0.997, 5, 0.01))
        d_dist, d_val = roots.get_distributions(), np.asarray(roots.get_values(), np.float32)
        assert [out[i]["visit_count_distributions"] for i in range(B)] == d_dist
        assert np.array_equal(np.asarray([out[i]["searched_value"] for i in range(B)], np.float32).view(np.uint32), d_val.view(np.uint32))
        for i in range(B):
            assert mask[i][out[i]["action"]] == 1.0 and sum(out[i]["visit_count_distributions"]) == S
            assert np.asarray(out[i]["predicted_value"]).shape == (1,)
        assert np.array_equal(np.asarray([out[i]["predicted_value"][0] for i in range(B)], np.float32), native.value)
    # network values against the torch restatement: the bounds of tests/parity_record.py
    with torch.no_grad():
        o = ref.initial_inference(obs_h)
        rv = tm.InverseScalarTransform()(o.value).reshape(-1).numpy()

    def rel(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.max(np.abs(a - b) / (1 + np.abs(b))))
    parity_record.check("reference_forward_collect/ez_atari96/B%d" % B,
                        dict(policy=rel([out[i]["predicted_policy_logits"] for i in range(B)], o.policy_logits.numpy()),
                             value=rel([out[i]["predicted_value"][0] for i in range(B)], rv)), extra=dict(batch=B))
    # eval forward: no noise, arg-max action, reproducible
    ev1 = pol._forward_eval(obs, action_mask=mask, to_play=to_play)
    ev2 = pol._forward_eval(obs, action_mask=mask, to_play=to_play)
    for i in range(B):
        assert ev1[i]["visit_count_distributions"] == ev2[i]["visit_count_distributions"] and ev1[i]["action"] == ev2[i]["action"]
        legal_i = np.nonzero(mask[i])[0]
        assert ev1[i]["action"] == legal_i[int(np.argmax(ev1[i]["visit_count_distributions"]))]


def test_python_recurrent_inference_and_the_foreign_loop_on_an_engine_model():
    """model.recurrent_inference(latent, hidden, action) with arrays (the reference's signature) against the torch restatement, and
    the reference-style search loop driven through it (arrays over PCIe every simulation) against the fused loop: same tree"""
    from oracle import torch_models as tm
    from lightzero_amd import _lib as L
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    from lightzero_amd.mcts.tree_search.mcts_ctree import EfficientZeroMCTSCtree
    B, A, S = 19, 6, 12
    ref = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A), seed=74)
    model = EfficientZeroModel(action_space_size=A).load_state_dict(ref.state_dict())
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(75))
    with torch.no_grad():
        o0 = ref.initial_inference(obs)
        act = torch.from_numpy(np.random.default_rng(1).integers(0, A, size=B))
        hid = (0.1 * torch.randn(1, B, 512, generator=torch.Generator().manual_seed(2)), 0.1 * torch.randn(1, B, 512, generator=torch.Generator().manual_seed(3)))
        o1 = ref.recurrent_inference(o0.latent_state, hid, act)
    d1 = model.recurrent_inference(o0.latent_state, hid, act)

    def rel(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        assert a.shape == b.shape, (a.shape, b.shape)
        return float(np.max(np.abs(a - b) / (1 + np.abs(b))))
    parity_record.check("python_recurrent_inference/ez_atari96/B%d" % B,
                        dict(latent=rel(d1.latent_state, o1.latent_state), policy=rel(d1.policy_logits, o1.policy_logits),
                             logits=max(rel(d1.value, o1.value), rel(d1.value_prefix, o1.value_prefix)),
                             h=rel(d1.reward_hidden_state[0], o1.reward_hidden_state[0]), c=rel(d1.reward_hidden_state[1], o1.reward_hidden_state[1])),
                        extra=dict(batch=B))
    # the foreign loop through the engine model's Python recurrent_inference vs the fused loop
    legal = [list(range(A))] * B
    noises = np.random.default_rng(4).dirichlet([0.3] * A, size=B).astype(np.float32)
    cfg = dict(num_simulations=S, pb_c_base=19652, pb_c_init=1.25, discount_factor=0.997, value_delta_max=0.01, lstm_horizon_len=5)
    fused = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
    fused.set_tiebreak(0)
    out = model.initial_inference(obs.cuda().contiguous(), fused)
    fused.prepare_from_inference(0.25, noises, [-1] * B)
    EfficientZeroMCTSCtree(cfg).search(fused, model, out.latent_state, out.reward_hidden_state, [-1] * B)
    lat0 = np.zeros((B, 64, 6, 6), np.float32)
    L.check(L.lib().lz_roots_read_latent(fused._h, 0, lat0.reshape(-1)))
    loop = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=model.engine)
    loop.set_tiebreak(0)
    loop.prepare(0.25, noises.tolist(), [0.] * B, out.policy_logits.tolist(), [-1] * B)
    EfficientZeroMCTSCtree(cfg).search(loop, model, lat0, (np.zeros((1, B, 512), np.float32), np.zeros((1, B, 512), np.float32)), [-1] * B)
    same = sum(int(a == b) for a, b in zip(loop.get_distributions(), fused.get_distributions()))
    # the loop's scalars come from torch's h^-1 of the device logits, the fused loop's from the device h^-1: a rare tie may flip
    assert same >= B - 1, "only %d / %d roots identical" % (same, B)


def test_mismatched_roots_and_model_raise_instead_of_falling_through():
    from oracle import torch_models as tm
    from lightzero_amd import _lib as L
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    from lightzero_amd.mcts.tree_search.mcts_ctree import EfficientZeroMCTSCtree
    B, A, S = 4, 6, 5
    sd = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A), seed=76).state_dict()
    m1 = EfficientZeroModel(action_space_size=A).load_state_dict(sd)
    m2 = EfficientZeroModel(action_space_size=A).load_state_dict(sd)
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(77)).cuda().contiguous()
    cfg = dict(num_simulations=S, pb_c_base=19652, pb_c_init=1.25, discount_factor=0.997, value_delta_max=0.01, lstm_horizon_len=5)
    roots = ez_tree.Roots(B, [list(range(A))] * B, action_space_size=A, max_simulations=S, engine=m1.engine)
    out = m1.initial_inference(obs, roots)
    roots.prepare_from_inference_no_noise([-1] * B)
    with pytest.raises(L.LzError, match="another model object"):
        EfficientZeroMCTSCtree(cfg).search(roots, m2, out.latent_state, out.reward_hidden_state, [-1] * B)
    tok = m1.initial_inference(obs)
    other = ez_tree.Roots(B, [list(range(A))] * B, action_space_size=A, max_simulations=S, engine=m1.engine)
    with pytest.raises(L.LzError, match="prepare"):
        EfficientZeroMCTSCtree(cfg).search(other, m1, tok.latent_state, tok.reward_hidden_state, [-1] * B)
    with pytest.raises(L.LzError, match="another model"):
        EfficientZeroMCTSCtree(cfg).search(other, m2, tok.latent_state, tok.reward_hidden_state, [-1] * B)
    with pytest.raises(L.LzError):
        m1.recurrent_inference(tok.latent_state, tok.reward_hidden_state, np.zeros(B, np.int64))
