"""CPU: pins the restated search drivers (oracle/search.py) to the reference's OWN drivers, and shows the drop-in claim of
INTEGRATION.md section 1 on the reference's unmodified code: /root/reference/lzero/mcts/tree_search/mcts_ctree.py is imported as
it lies (tests/ref_driver_loader.py: the reference's compiled ctree, its real scaling_transform.py, stub easydict) and

  * EfficientZeroMCTSCtree.search / MuZeroMCTSCtree.search / GumbelMuZeroMCTSCtree.search give visit distributions identical and root
    values BIT-EQUAL to oracle/search.py's ez_search / mz_search / gmz_search on the same tree module and model;
  * the same reference driver with ANOTHER module behind the Cython surface (the C restatement oracle/ctree.py here; the HBM trees
    of lightzero_amd on the GPU box, tests/test_reference_driver_gpu.py) gives the same search;
  * the recorded runs committed as tests/golden/driver_*.npz are what the reference produces today."""
import os
import sys

import numpy as np
import pytest
import torch

import ref_driver_loader as rdl
from oracle import ctree as octree, search as osearch, torch_models as tm

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
import make_golden_driver as mgd  # noqa: E402

needs_ref = pytest.mark.skipif(not rdl.available(), reason="/root/reference is not on this machine")


def _cfg_dict(cfg):
    return dict(num_simulations=cfg.num_simulations, pb_c_base=cfg.pb_c_base, pb_c_init=cfg.pb_c_init, discount_factor=cfg.discount_factor,
                value_delta_max=cfg.value_delta_max, lstm_horizon_len=cfg.lstm_horizon_len, root_noise_weight=cfg.root_noise_weight,
                support_range=tuple(cfg.model.value_support_range))


@needs_ref
@pytest.mark.parametrize("B,S", [(16, 20), (5, 33)])
def test_reference_efficientzero_driver_equals_restated_driver(B, S):
    ns = rdl.load()
    A = 6
    ref_model = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A), seed=B)
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(S))
    rng = np.random.default_rng(B + S)
    legal = [np.nonzero(np.r_[True, rng.random(A - 1) < 0.8])[0].tolist() for _ in range(B)]
    noises = [rng.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal]
    cfg = rdl.driver_cfg(S)
    with torch.no_grad():
        out = ref_model.initial_inference(obs)
    lat = out.latent_state.numpy(); rh = (out.reward_hidden_state[0].numpy(), out.reward_hidden_state[1].numpy())
    logits = out.policy_logits.numpy().tolist()
    res = []
    for which in ("reference", "restated"):
        roots = ns.ez_tree.Roots(B, legal)
        roots.prepare(0.25, noises, [0.] * B, logits, [-1] * B)
        if which == "reference":
            ns.driver.EfficientZeroMCTSCtree(cfg).search(roots, rdl.MutableOutputModel(ref_model), lat, (rh[0].copy(), rh[1].copy()), [-1] * B)
        else:
            osearch.ez_search(ns.ez_tree, roots, ref_model, lat, (rh[0].copy(), rh[1].copy()), [-1] * B, _cfg_dict(cfg))
        res.append((roots.get_distributions(), np.asarray(roots.get_values(), np.float32)))
    assert res[0][0] == res[1][0]
    assert np.array_equal(res[0][1].view(np.uint32), res[1][1].view(np.uint32))
    assert all(sum(d) == S for d in res[0][0])


@needs_ref
def test_reference_muzero_driver_equals_restated_driver_two_player():
    ns = rdl.load()
    B, A, S = 10, 36, 30
    kw = dict(observation_shape=(3, 6, 6), action_space_size=A, downsample=False, num_channels=32,
              reward_support_range=(-10., 11., 1.), value_support_range=(-10., 11., 1.))
    ref_model = tm.synthetic_init(tm.MuZeroModel(**kw), seed=3)
    rng = np.random.default_rng(9)
    obs = torch.from_numpy((rng.random((B, 3, 6, 6)) < 0.3).astype(np.float32))
    legal = [np.nonzero(np.r_[True, rng.random(A - 1) < 0.7])[0].tolist() for _ in range(B)]
    to_play = rng.integers(1, 3, size=B).tolist()
    noises = [rng.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal]
    cfg = rdl.driver_cfg(S, discount_factor=1.0, env_type="board_games", support=(-10., 11., 1.))
    with torch.no_grad():
        out = ref_model.initial_inference(obs)
    lat = out.latent_state.numpy(); logits = out.policy_logits.numpy().tolist()
    res = []
    for which in ("reference", "restated"):
        roots = ns.mz_tree.Roots(B, legal)
        roots.prepare(0.25, noises, [0.] * B, logits, list(to_play))
        if which == "reference":
            ns.driver.MuZeroMCTSCtree(cfg).search(roots, rdl.MutableOutputModel(ref_model), lat, list(to_play))
        else:
            osearch.mz_search(ns.mz_tree, roots, ref_model, lat, list(to_play), _cfg_dict(cfg))
        res.append((roots.get_distributions(), np.asarray(roots.get_values(), np.float32)))
    assert res[0][0] == res[1][0]
    assert np.array_equal(res[0][1].view(np.uint32), res[1][1].view(np.uint32))


@needs_ref
def test_reference_gumbel_driver_equals_restated_driver():
    ns = rdl.load()
    B, A, S, m = 12, 6, 24, 4
    ref_model = tm.synthetic_init(tm.MuZeroModel(action_space_size=A), seed=5)
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(6))
    rng = np.random.default_rng(7)
    legal = [np.nonzero(np.r_[True, rng.random(A - 1) < 0.8])[0].tolist() for _ in range(B)]
    noises = [rng.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal]
    cfg = rdl.driver_cfg(S, max_num_considered_actions=m)
    ist = tm.InverseScalarTransform()
    with torch.no_grad():
        out = ref_model.initial_inference(obs)
    lat = out.latent_state.numpy(); logits = out.policy_logits.numpy().tolist()
    pred = ist(out.value).reshape(-1).numpy().tolist()
    res = []
    for which in ("reference", "restated"):
        roots = ns.gmz_tree.Roots(B, legal)
        roots.prepare(0.25, noises, [0.] * B, pred, logits, [-1] * B)
        if which == "reference":
            ns.driver.GumbelMuZeroMCTSCtree(cfg).search(roots, rdl.MutableOutputModel(ref_model), lat, [-1] * B)
        else:
            osearch.gmz_search(ns.gmz_tree, roots, ref_model, lat, [-1] * B, dict(_cfg_dict(cfg), max_num_considered_actions=m))
        res.append((roots.get_distributions(), np.asarray(roots.get_values(), np.float32), np.asarray(roots.get_policies(0.997, A), np.float32)))
    assert res[0][0] == res[1][0]
    assert np.array_equal(res[0][1].view(np.uint32), res[1][1].view(np.uint32))
    assert np.array_equal(res[0][2].view(np.uint32), res[1][2].view(np.uint32))


@needs_ref
@pytest.mark.parametrize("name", sorted(mgd.CASES))
def test_reference_driver_golden_is_current_and_tree_module_is_swappable(name):
    """(a) the committed recording equals a fresh run of the reference driver on the reference tree; (b) the reference driver
    UNMODIFIED on another implementation of the Cython surface (oracle/ctree.py) reproduces it: the module swap of INTEGRATION.md"""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    case = mgd.CASES[name]
    fresh = mgd.run_reference(case)
    swapped = mgd.run_reference(case, tree_override=octree.ez_tree if case["family"] == "ez" else octree.mz_tree)

    def same(a, b):
        return np.array_equal(a.view(np.uint32), b.view(np.uint32)) if a.dtype == np.float32 else np.array_equal(a, b)
    for k in g.files:   # (b): same machine, same model arithmetic -> the two trees must give the same search, bit for bit
        assert same(fresh[k], swapped[k]), "reference driver on the C restatement: %s differs from the run on the reference's tree" % k
    # (a): the golden was recorded with THIS torch build on the machine that made it; another CPU's softmax may differ in the last
    # bits of the network scalars, after which the searches legitimately part ways
    if not (same(fresh["root_pred"], g["root_pred"]) and same(fresh["value"][0], g["value"][0])):
        pytest.skip("this machine's torch CPU arithmetic differs in the last bits from the one that recorded the golden")
    for k in g.files:
        assert same(g[k], fresh[k]), "fresh reference run: %s differs from the golden" % k


@pytest.mark.parametrize("name", sorted(mgd.CASES))
def test_restated_tree_replays_driver_golden(name):
    """runs everywhere: the recorded raw outputs of the golden through the restated driver loop on oracle/ctree.py"""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    case = mgd.CASES[name]
    dist, values = replay_golden(case, g, octree.ez_tree if case["family"] == "ez" else octree.mz_tree,
                                 roots_kwargs=dict(action_space_size=case["A"], max_simulations=case["S"]))
    assert np.array_equal(dist, g["distributions"])
    assert np.array_equal(values.view(np.uint32), g["values"].view(np.uint32))


def replay_golden(case, g, tree, roots_kwargs=None, search=None):
    """drive `tree` with the golden's recording through a driver loop (default: oracle/search.py's) by way of a model that returns
    the recorded policy logits and CHECKS the selections it is called with, and of inverse_scalar_transform handles that return the
    recorded post-transform scalars.  ``search(roots, model, handle, lat0, hc0, to_play, cfg)`` runs another driver."""
    _, legal, to_play, noises = mgd.inputs(case)
    fam, B, A, S = case["family"], case["B"], case["A"], case["S"]
    support = tuple(case["kw"].get("value_support_range", (-300., 301., 1.)))
    model = ReplayModel(g, fam)
    handle = ReplayHandle(g)
    roots = tree.Roots(B, legal, **(roots_kwargs or {}))
    roots.prepare(0.25, noises, [0.] * B, g["root_logits"].tolist(), list(to_play))
    cfg = dict(num_simulations=S, pb_c_base=19652, pb_c_init=1.25, discount_factor=1.0 if case.get("two_player") else 0.997,
               value_delta_max=0.01, lstm_horizon_len=5, root_noise_weight=0.25, support_range=support,
               env_type="board_games" if case.get("two_player") else "not_board_games",
               model=dict(value_support_range=support, reward_support_range=support, categorical_distribution=True))
    lat0 = model.latent(0, B)
    hc0 = (np.zeros((1, B, 2), np.float32), np.zeros((1, B, 2), np.float32))
    if search is not None:
        search(roots, model, handle, lat0, hc0, list(to_play), cfg)
    elif fam == "ez":
        osearch.ez_search(tree, roots, model, lat0, hc0, list(to_play), cfg, ist=handle)
    else:
        osearch.mz_search(tree, roots, model, lat0, list(to_play), cfg, ist=handle)
    assert model.calls == S and handle.calls == 2 * S
    dist = np.full((B, A), -1, np.int32)
    for i, d in enumerate(roots.get_distributions()):
        dist[i, :len(d)] = d
    return dist, np.asarray(roots.get_values(), np.float32)


class ReplayHandle(object):
    """stands in for value_/reward_inverse_scalar_transform_handle: the model's value / value-prefix outputs are tags
    [B, 2] = (simulation, kind); the handle returns what the reference's handle returned there"""

    def __init__(self, g):
        self.g, self.calls = g, 0

    def __call__(self, tagged):
        t = tagged.detach().cpu().numpy()
        s, kind = int(t[0, 0]), int(t[0, 1])
        assert (t[:, 0] == s).all() and (t[:, 1] == kind).all()
        self.calls += 1
        return torch.from_numpy(np.ascontiguousarray(self.g["vp" if kind else "value"][s]).reshape(-1, 1).copy())


class ReplayModel(object):
    """recurrent_inference returns the golden's recording of simulation s (policy logits; tags for the two scalars) and asserts that
    the driver + tree under test selected what the reference selected: the action of every root and -- through the latent, which
    encodes (slot, root) -- the (parent slot, batch index) gather"""

    def __init__(self, g, family):
        self.g, self.family, self.calls = g, family, 0

    def eval(self):
        return self

    @staticmethod
    def latent(slot, B):
        return np.stack([np.full(B, slot, np.float32), np.arange(B, dtype=np.float32)], 1)

    def recurrent_inference(self, latent, *rest):
        g = self.g
        action = rest[-1]
        s = self.calls
        B = latent.shape[0]
        lat = latent.detach().cpu().numpy()
        assert np.array_equal(lat[:, 0].astype(np.int32), g["ix"][s]), "simulation %d: parent slots differ from the reference's" % s
        assert np.array_equal(lat[:, 1].astype(np.int32), np.arange(B)), "simulation %d: batch indices of the gather" % s
        assert np.array_equal(action.detach().cpu().numpy().reshape(-1), g["action"][s]), "simulation %d: actions differ from the reference's" % s
        self.calls += 1
        nxt = torch.from_numpy(self.latent(s + 1, B))
        tag = lambda kind: torch.from_numpy(np.stack([np.full(B, s, np.float32), np.full(B, kind, np.float32)], 1))  # noqa: E731
        pol = torch.from_numpy(np.ascontiguousarray(g["policy"][s]))
        if self.family == "ez":
            return tm.EZNetworkOutput(tag(0), tag(1), pol, nxt, (torch.zeros(1, B, 2), torch.zeros(1, B, 2)))
        return tm.MZNetworkOutput(tag(0), tag(1), pol, nxt)
