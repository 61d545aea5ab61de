"""Generates tests/golden/driver_*.npz: the reference's OWN search drivers (lzero/mcts/tree_search/mcts_ctree.py, imported as it
lies by tests/ref_driver_loader.py) running on the reference's own compiled ctree (oracle/_ref/det) with the torch model of
oracle/torch_models.py (pinned bit-equal to the reference's model modules), recorded per simulation:

    ix / action / search_len                   what batch_traverse returned
    value / vp (value prefix | reward)         what the driver's inverse_scalar_transform handles returned for that simulation
    policy                                     the policy logits
and at the end  distributions, values          Roots.get_distributions() / get_values()

The scalars are recorded AFTER the handles, not as support-wide logits: torch's CPU softmax / sum differ in the last bits between
machines (vector width, thread partition), and one such bit decides a tie somewhere in a search -- a golden of raw logits would
only replay on the machine that made it.  A machine without /root/reference replays the recording through a tree implementation
behind the same driver loop with handles that return the recorded scalars, and must arrive at the same distributions / values
(tests/test_reference_driver_gpu.py: lightzero_amd's HBM trees driven by the foreign-model loop of
lightzero_amd/mcts/tree_search/mcts_ctree.py).  h^-1 itself is pinned in tests/test_torch_models_vs_reference.py.

    python tests/golden/make_golden_driver.py [case ...]      (needs /root/reference; run from the repository root)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.dirname(HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)

CASES = {
    # name: family, batch, actions, simulations, model kwargs, two-player
    "driver_ez_atari_b24": dict(family="ez", B=24, A=6, S=50, kw=dict(observation_shape=(4, 96, 96), action_space_size=6), seed=61),
    "driver_mz_gomoku_2p_b16": dict(family="mz", B=16, A=36, S=40, seed=62, two_player=True,
                                    kw=dict(observation_shape=(3, 6, 6), action_space_size=36, downsample=False, num_channels=32,
                                            reward_support_range=(-10., 11., 1.), value_support_range=(-10., 11., 1.))),
    # the Chinese chess preset (zoo/board_games/chinese_chess/config/chinese_chess_muzero_bot_mode_config.py:31-46): 2086 moves, a few
    # dozen legal at a root -- the action space beyond 256 of csrc/lz_tree_wide.hip behind the reference's own MuZero loop
    "driver_mz_xiangqi_2p_b4": dict(family="mz", B=4, A=2086, S=30, seed=63, two_player=True, legal_p=0.02,
                                    kw=dict(observation_shape=(57, 10, 9), action_space_size=2086, num_res_blocks=6,
                                            num_channels=128, reward_head_hidden_channels=[128], value_head_hidden_channels=[128],
                                            policy_head_hidden_channels=[256], downsample=False,
                                            reward_support_range=(-1., 1., 1.), value_support_range=(-1., 1., 1.),
                                            discrete_action_encoding_type='not_one_hot')),
}


def inputs(case):
    rng = np.random.default_rng(case["seed"])
    B, A = case["B"], case["A"]
    obs = rng.random((B,) + tuple(case["kw"]["observation_shape"]), dtype=np.float32)
    legal = []
    for _ in range(B):
        m = rng.random(A) < case.get("legal_p", 0.8)
        m[rng.integers(0, A)] = True
        legal.append(np.nonzero(m)[0].tolist())
    to_play = rng.integers(1, 3, size=B).tolist() if case.get("two_player") else [-1] * B
    noises = [rng.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal]
    return obs, legal, to_play, noises


def run_reference(case, tree_override=None):
    """-> dict of arrays (the golden's content).  tree_override: a module with the ctree surface to run the reference driver on
    instead of the reference's compiled tree (the driver module's global is swapped for the call)."""
    import ref_driver_loader as rdl
    from oracle import torch_models as tm
    ns = rdl.load()
    assert ns is not None, "needs /root/reference and its compiled ctree"
    fam, B, A, S = case["family"], case["B"], case["A"], case["S"]
    cls = tm.EfficientZeroModel if fam == "ez" else tm.MuZeroModel
    ref_model = tm.synthetic_init(cls(**case["kw"]), seed=case["seed"])
    record = []
    model = rdl.MutableOutputModel(ref_model, record)
    obs, legal, to_play, noises = inputs(case)
    support = tuple(case["kw"].get("value_support_range", (-300., 301., 1.)))
    cfg = rdl.driver_cfg(S, discount_factor=1.0 if case.get("two_player") else 0.997,
                         env_type="board_games" if case.get("two_player") else "not_board_games", support=support)
    drv = ns.driver
    gname = "tree_efficientzero" if fam == "ez" else "tree_muzero"
    tree = tree_override if tree_override is not None else (ns.ez_tree if fam == "ez" else ns.mz_tree)
    saved = getattr(drv, gname)
    setattr(drv, gname, tree)
    traverse_log, handle_log = [], []
    orig_traverse = tree.batch_traverse

    def logging_traverse(roots, *a, **k):
        res = orig_traverse(roots, *a, **k)
        results = a[4] if len(a) > 4 else k["results"]
        traverse_log.append((list(res[0]), list(res[2]), list(results.get_search_len())))
        return res

    class LoggingHandle(object):
        def __init__(self, inner, kind):
            self.inner, self.kind = inner, kind

        def __call__(self, logits):
            out = self.inner(logits)
            handle_log.append((self.kind, out.detach().cpu().numpy().reshape(-1).copy()))
            return out
    try:
        ist = drv.InverseScalarTransform(drv.DiscreteSupport(*support, "cpu"), True)
        with torch.no_grad():
            out = ref_model.initial_inference(torch.from_numpy(obs))
        root_logits = out.policy_logits.numpy()
        pred = ist(out.value).numpy().reshape(-1)
        roots = tree.Roots(B, legal) if tree_override is None else tree.Roots(B, legal, action_space_size=A, max_simulations=S)
        roots.prepare(cfg.root_noise_weight, noises, [0.] * B, root_logits.tolist(), to_play)
        mcts = drv.EfficientZeroMCTSCtree(cfg) if fam == "ez" else drv.MuZeroMCTSCtree(cfg)
        mcts.value_inverse_scalar_transform_handle = LoggingHandle(mcts.value_inverse_scalar_transform_handle, "value")
        mcts.reward_inverse_scalar_transform_handle = LoggingHandle(mcts.reward_inverse_scalar_transform_handle, "reward")
        tree.batch_traverse = logging_traverse
        if fam == "ez":
            mcts.search(roots, model, out.latent_state.numpy(), (out.reward_hidden_state[0].numpy(), out.reward_hidden_state[1].numpy()), to_play)
        else:
            mcts.search(roots, model, out.latent_state.numpy(), to_play)
    finally:
        tree.batch_traverse = orig_traverse
        setattr(drv, gname, saved)
    # the MuZero driver calls recurrent_inference twice per simulation (mcts_ctree.py:338-345) and keeps the second result
    per_sim = len(record) // S
    assert per_sim in (1, 2) and len(traverse_log) == S and len(handle_log) == 2 * S
    outs = [record[per_sim * s + per_sim - 1][1] for s in range(S)]
    # per simulation the drivers transform the value first, then the value prefix (mcts_ctree.py:839-841; through the VALUE handle
    # in the EfficientZero driver) / the reward (:347-348)
    assert all(handle_log[2 * s][0] == "value" for s in range(S))
    g = dict(root_logits=root_logits, root_pred=pred,
             ix=np.asarray([t[0] for t in traverse_log], np.int32), action=np.asarray([t[1] for t in traverse_log], np.int32),
             search_len=np.asarray([t[2] for t in traverse_log], np.int32),
             value=np.stack([handle_log[2 * s][1] for s in range(S)]), vp=np.stack([handle_log[2 * s + 1][1] for s in range(S)]),
             policy=np.stack([o.policy_logits.numpy() for o in outs]),
             values=np.asarray(roots.get_values(), np.float32))
    A_pad = np.full((B, A), -1, np.int32)
    for i, d in enumerate(roots.get_distributions()):
        A_pad[i, :len(d)] = d
    g["distributions"] = A_pad
    return g


if __name__ == "__main__":
    for name, case in CASES.items():
        if sys.argv[1:] and name not in sys.argv[1:]:
            continue
        g = run_reference(case)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **g)
        print(name, {k: v.shape for k, v in g.items()}, "visits", g["distributions"].clip(0).sum(1)[:4])
