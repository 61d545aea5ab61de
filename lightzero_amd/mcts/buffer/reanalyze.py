"""The reanalyze caller of the hot path: ``_compute_target_policy_reanalyzed`` of
lzero/mcts/buffer/game_buffer_efficientzero.py:325-468 (and game_buffer_muzero.py:575-720, the same function without the reward
hidden state) -- ``batch_size * (num_unroll_steps + 1)`` stored observations are searched again with the current target model and
the visit counts become the policy targets; the refreshed visit distributions / root values are written back into the game
segments' ``child_visit_segment`` / ``root_value_segment``.

Reference: per-transition Python lists (legal actions, one Dirichlet draw per root), the model in ``mini_infer_size`` slices (GPU
memory of the time), ``MCTSCtree.roots / prepare[_no_noise] / search``, then a doubly nested loop building the targets.  Here: ONE
initial inference + fused search over all roots (engine model: nothing returns to the host in between; 1536 roots x 50 simulations is
the batch the exact replay gate runs, tests/test_exact_replay_gpu.py), masks -> legal lists by one np.nonzero, the targets by one
normalisation and one scatter; only the write-back into the segments' Python containers stays a loop (that IS the reference's data
structure).  Any other model object goes through the same MCTS class's reference-shaped loop.

``policy_re_context`` is the reference's tuple (game_buffer_muzero.py:_prepare_policy_reanalyzed_context):
(policy_obs_list, policy_mask, pos_in_game_segment_list, batch_index_list, child_visits, root_values, game_segment_lens,
action_mask_segment, to_play_segment)."""
import os

import numpy as np


def _g(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def preprocess_to_play_and_action_mask(game_segment_batch_size, to_play_segment, action_mask_segment, pos_in_game_segment_list,
                                       unroll_steps, action_space_size, continuous_action_space=False):
    """game_buffer.py:480-523: the to_play / action mask of the ``unroll_steps + 1`` positions from each sampled position on; positions
    past the end of a segment get to_play -1 and an all-ones mask.  Returns (to_play [T] int64, action_mask [T, A] | None)."""
    U1 = int(unroll_steps) + 1
    T = int(game_segment_batch_size) * U1
    to_play = np.full(T, -1, np.int64)
    mask = None if continuous_action_space else np.ones((T, int(action_space_size)), np.int8)
    for bs in range(int(game_segment_batch_size)):
        p = int(pos_in_game_segment_list[bs])
        tp = np.asarray(to_play_segment[bs][p:p + U1]).reshape(-1)
        to_play[bs * U1:bs * U1 + len(tp)] = tp
        if mask is not None:
            am = np.asarray(action_mask_segment[bs][p:p + U1])
            if len(am):
                mask[bs * U1:bs * U1 + len(am)] = am.reshape(len(am), -1)
    return to_play, mask


def targets_from_search(policy_mask, distributions, counts, action_mask, action_space_size, fixed_action_space):
    """game_buffer_efficientzero.py:410-465 for all positions at once: ``distributions`` [T, >= max(counts)] visit counts in
    legal-action order, ``counts`` [T] number of legal actions.  Returns (targets [T, A] float64, normalised visit lists' flat form is
    the same numbers: targets of a fixed action space ARE the child visits)."""
    T, A = len(policy_mask), int(action_space_size)
    dist = np.asarray(distributions, np.float64)
    counts = np.asarray(counts, np.int64)
    valid = np.arange(dist.shape[1])[None, :] < counts[:, None]
    dist = np.where(valid, dist, 0.0)
    prob = dist / dist.sum(1, keepdims=True)          # visit_count / sum_visits, binary64 like the reference's Python floats
    targets = np.zeros((T, A), np.float64)
    if fixed_action_space:
        targets[:, :] = prob[:, :A]
    else:   # two-player board games: scatter the legal-order policy to action indices (one np.nonzero: row-major = legal order)
        rows, cols = np.nonzero(np.asarray(action_mask) != 0)
        pos = np.arange(len(rows)) - np.repeat(np.cumsum(counts) - counts, counts)
        targets[rows, cols] = prob[rows, pos]
    targets[np.asarray(policy_mask) == 0] = 0.0       # invalid padding positions: zeros, so that their cross entropy is 0 (:414-416)
    return targets, prob


def compute_target_policy_reanalyzed(policy_re_context, model, cfg, mcts_cls=None, search_results=None):
    """-> ``batch_target_policies_re`` [game_segment_batch_size, num_unroll_steps + 1, action_space_size] (np.ndarray, like the
    reference's ``np.array`` of nested lists); writes the refreshed child visits / root values into ``child_visits`` / ``root_values``.
    ``cfg``: the policy config (num_unroll_steps, model.action_space_size, model.model_type, root_dirichlet_alpha, root_noise_weight,
    reanalyze_noise, action_type, device ...).  ``mcts_cls``: the MCTS class (default: by model family).  ``search_results``: testing
    hook -- a callable (obs, legal_actions, to_play, noises) -> (distributions [T, A], counts [T], values [T]) that replaces the
    inference + search."""
    if policy_re_context is None:
        return []
    (policy_obs_list, policy_mask, pos_in_game_segment_list, batch_index_list, child_visits, root_values, game_segment_lens,
     action_mask_segment, to_play_segment) = policy_re_context
    mcfg = _g(cfg, "model", {})
    A, U = int(_g(mcfg, "action_space_size")), int(_g(cfg, "num_unroll_steps"))
    T, B = len(policy_obs_list), len(pos_in_game_segment_list)
    to_play, action_mask = preprocess_to_play_and_action_mask(B, to_play_segment, action_mask_segment, pos_in_game_segment_list, U, A,
                                                              bool(_g(mcfg, "continuous_action_space", False)))
    if action_mask is None or bool(_g(mcfg, "continuous_action_space", False)):
        # game_buffer_sampled_efficientzero.py builds K sampled actions per root instead of legal-action lists: not this function
        raise NotImplementedError("compute_target_policy_reanalyzed serves discrete action spaces (game_buffer_efficientzero.py / "
                                  "game_buffer_muzero.py); continuous (Sampled EfficientZero) reanalysis is not built")
    m2 = action_mask != 0
    counts = m2.sum(1)
    flat = np.nonzero(m2)[1].tolist()
    ends = np.cumsum(counts).tolist()
    legal_actions = [flat[a:b] for a, b in zip([0] + ends[:-1], ends)]
    obs = np.asarray(policy_obs_list)
    if obs.ndim == 5:      # prepare_observation 'conv' (lzero/mcts/utils.py:104-111): [T, S, C, W, H] -> [T, S C, W, H]
        obs = obs.reshape(T, obs.shape[1] * obs.shape[2], obs.shape[3], obs.shape[4])
    elif obs.ndim == 3 and str(_g(mcfg, "model_type", "conv")).startswith("mlp"):
        obs = obs.reshape(T, -1)
    noises = None
    # RNG-stream note: the reference draws these Dirichlet noises unconditionally (game_buffer_efficientzero.py:377-380) and uses them
    # only when reanalyze_noise is set; here they are drawn only when used, so with reanalyze_noise = False a seeded np.random stream is
    # NOT advanced by this call (it is by the reference's).  Set LZ_REANALYZE_DRAW_ALWAYS=1 for stream parity with the reference.
    if not bool(_g(cfg, "reanalyze_noise", False)) and os.environ.get("LZ_REANALYZE_DRAW_ALWAYS"):
        np.random.dirichlet([float(_g(cfg, "root_dirichlet_alpha"))] * A, size=T)
    if bool(_g(cfg, "reanalyze_noise", False)):
        # game_buffer_efficientzero.py:377-380: one Dirichlet over the whole action space per root; a root uses its first #legal entries
        full = np.random.dirichlet([float(_g(cfg, "root_dirichlet_alpha"))] * A, size=T).astype(np.float32)
        noises = [full[i, :counts[i]] for i in range(T)]
    if search_results is not None:
        dist, cnt, values = search_results(obs, legal_actions, to_play.tolist(), noises)
    else:
        import torch
        if not getattr(model, "_is_lz_engine_model", False):
            raise NotImplementedError("compute_target_policy_reanalyzed drives engine models; for any other model run the reference's own "
                                      "function with lightzero_amd's MCTSCtree in place of lzero's (INTEGRATION.md)")
        if mcts_cls is None:
            from ..tree_search.mcts_ctree import EfficientZeroMCTSCtree, MuZeroMCTSCtree
            mcts_cls = EfficientZeroMCTSCtree if getattr(model, "_uses_lstm", True) else MuZeroMCTSCtree
        S = int(_g(cfg, "num_simulations"))
        roots = mcts_cls.roots(T, legal_actions, action_space_size=A, max_simulations=S)
        seed = _g(cfg, "mcts_seed", None)
        if _g(cfg, "mcts_tiebreak", None) is not None or seed is not None:
            roots.set_tiebreak(0 if _g(cfg, "mcts_tiebreak", "random") == "first" else 1, seed=seed)
        data = obs if hasattr(obs, "data_ptr") else torch.from_numpy(np.ascontiguousarray(obs, np.float32)).cuda()
        model.initial_inference(data, roots, fetch=False)      # latent / LSTM state stay in HBM; nothing is read back before the search
        tp = to_play.tolist()
        if noises is not None:
            roots.prepare_from_inference(float(_g(cfg, "root_noise_weight")), noises, tp)
        else:
            roots.prepare_from_inference_no_noise(tp)
        tokens = ("hbm-pool", roots)
        mcts = mcts_cls(cfg)
        if getattr(model, "_uses_lstm", True):
            mcts.search(roots, model, tokens, tokens, tp)
        else:
            mcts.search(roots, model, tokens, tp)
        dist, cnt, values = roots.get_search_results()[:3]
    fixed = str(_g(cfg, "action_type", "fixed_action_space")) == "fixed_action_space"
    targets, prob = targets_from_search(policy_mask, dist, cnt, action_mask, A, fixed)
    # ---- write-back into the segments (:427-433): the reference's containers, one assignment per valid position
    U1 = U + 1
    pm = np.asarray(policy_mask)
    values = np.asarray(values)
    for b, (state_index, child_visit, root_value) in enumerate(zip(pos_in_game_segment_list, child_visits, root_values)):
        for k in range(U1):
            i = b * U1 + k
            if pm[i] != 0:
                child_visit[state_index + k] = prob[i, :cnt[i]].tolist()
                root_value[state_index + k] = values[i]
    return targets.reshape(B, U1, A)
