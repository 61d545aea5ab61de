"""Collect / eval halves of lzero/policy/efficientzero.py::EfficientZeroPolicy on the MI355X engine.

``_forward_collect(data, action_mask, temperature, to_play, epsilon, ready_env_id)`` (efficientzero.py:539-657)
and ``_forward_eval(data, action_mask, to_play, ready_env_id)`` (:670-747) keep the reference's arguments and
the per-env output dict (``action``, ``visit_count_distributions``, ``visit_count_distribution_entropy``,
``searched_value``, ``predicted_value``, ``predicted_policy_logits``).  The learn half is out of scope.
"""
import numpy as np

from .. import _lib as L
from ..mcts.tree_search.mcts_ctree import EfficientZeroMCTSCtree as MCTSCtree
from .utils import CheckpointIngest, select_action


def _g(cfg, name, default=None):
    if isinstance(cfg, dict):
        return cfg.get(name, default)
    return getattr(cfg, name, default)


def _mcts_seed(cfg):
    """``mcts_seed`` (optional policy config entry): pins the device-side random streams of the search (stochastic tie-breaks,
    sampled actions) for reproducible runs; mixed with the rank so that data-parallel collectors still differ.  None: the
    roots draw their seed from np.random (see lightzero_amd._lib.process_seed)."""
    import os
    seed = _g(cfg, "mcts_seed", None)
    if seed is None:
        return None
    rank = int(os.environ.get("RANK", os.environ.get("LOCAL_RANK", "0")))
    return (int(seed) ^ ((rank + 1) * 0x9E3779B97F4A7C15)) & (2 ** 63 - 1)


class _HbmTokens(object):
    """stands in for the network output of an engine model whose latent / LSTM state stayed in HBM"""

    def __init__(self, roots):
        self.latent_state = ("hbm-pool", roots)
        self.reward_hidden_state = ("hbm-pool", roots)


class EfficientZeroPolicy(CheckpointIngest):
    def __init__(self, cfg, model):
        """cfg: the reference policy config (dict / EasyDict-like): num_simulations, discount_factor,
        lstm_horizon_len, root_dirichlet_alpha, root_noise_weight, pb_c_base, pb_c_init, value_delta_max, ...
        model: lightzero_amd.model.efficientzero_model.EfficientZeroModel with weights loaded."""
        self._cfg = cfg
        self._collect_model = model
        self._eval_model = model
        mcfg = dict(num_simulations=_g(cfg, "num_simulations", 50), discount_factor=_g(cfg, "discount_factor", 0.997),
                    lstm_horizon_len=_g(cfg, "lstm_horizon_len", 5), pb_c_base=_g(cfg, "pb_c_base", 19652),
                    pb_c_init=_g(cfg, "pb_c_init", 1.25), value_delta_max=_g(cfg, "value_delta_max", 0.01),
                    root_dirichlet_alpha=_g(cfg, "root_dirichlet_alpha", 0.3),
                    root_noise_weight=_g(cfg, "root_noise_weight", 0.25), env_type=_g(cfg, "env_type", "not_board_games"),
                    model=_g(cfg, "model", {}) or {})
        self._mcfg = mcfg
        self._mcts_collect = MCTSCtree(mcfg)
        self._mcts_eval = MCTSCtree(mcfg)
        self._collect_mcts_temperature = 1.
        self.collect_epsilon = 0.0
        self._roots_cache = {}
        # "random": the reference's stochastic tie rule (rand() over the tie list, cnode.cpp:691);
        # "first": deterministic first arg-max (parity / reproducible evaluation)
        self._tiebreak = {"random": 1, "first": 0}[_g(cfg, "mcts_tiebreak", "random")]
        # True: select_action (temperature sampling / arg-max + entropy) runs as one device kernel over all roots
        # (lz_roots_select_action) instead of the reference's per-env Python loop with np.random.choice
        self._device_select = bool(_g(cfg, "device_select_action", False))
        # True: the Dirichlet exploration noise of a collect forward is drawn on the device (lz_roots_prepare_from_inference_dirichlet)
        # instead of with np.random.dirichlet per env (efficientzero.py:599-602): same distribution, the engine's own random stream
        self._device_noise = bool(_g(cfg, "device_root_noise", False))

    def forward(self, *args, **kwargs):
        return self._forward_collect(*args, **kwargs)

    def _roots(self, n, legal_actions):
        # the reference builds a fresh Roots per forward (efficientzero.py:605); here the HBM pools of a batch size
        # are allocated once and re-armed with the new legal-action lists
        roots = self._roots_cache.get(n)
        if roots is None:
            roots = MCTSCtree.roots(n, legal_actions, action_space_size=self._collect_model.action_space_size,
                                    max_simulations=int(self._mcfg["num_simulations"]))
            roots.set_tiebreak(self._tiebreak, seed=_mcts_seed(self._cfg))
            self._roots_cache[n] = roots
        else:
            roots.reset(legal_actions)
        return roots

    def _search(self, mcts, roots, model, network_output, to_play):
        mcts.search(roots, model, network_output.latent_state, network_output.reward_hidden_state, to_play)

    def _forward_collect(self, data, action_mask=None, temperature=1, to_play=[-1], epsilon=0.25, ready_env_id=None,
                         **kwargs):
        self._collect_mcts_temperature = temperature
        self.collect_epsilon = epsilon
        active_collect_env_num = data.shape[0]
        if ready_env_id is None:
            ready_env_id = np.arange(active_collect_env_num)
        output = {i: None for i in ready_env_id}
        to_play = list(to_play) if len(to_play) == active_collect_env_num else [to_play[0]] * active_collect_env_num
        # engine model + cached roots: the representation network is launched first, and the host work below (legal lists,
        # re-arming the roots, noise draws) runs while it computes
        early = None
        if getattr(self._collect_model, "_is_lz_engine_model", False):
            early = self._roots_cache.get(active_collect_env_num)
            if early is not None and hasattr(early, "get_search_results"):
                self._collect_model.initial_inference(data, early, fetch=False)
            else:
                early = None
        # efficientzero.py:595's legal lists from ONE np.nonzero over the [B, A] mask
        mask2d = np.asarray(action_mask) != 0
        if mask2d.ndim == 2 and mask2d.shape[0] == active_collect_env_num:
            flat = np.nonzero(mask2d)[1].tolist()
            ends = np.cumsum(mask2d.sum(1)).tolist()
            legal_actions = [flat[a:b] for a, b in zip([0] + ends[:-1], ends)]
        else:
            mask2d = None
            legal_actions = [np.nonzero(action_mask[j])[0].tolist() for j in range(active_collect_env_num)]
        if early is None:
            roots = self._roots(active_collect_env_num, legal_actions)
        elif mask2d is not None and all(legal_actions) and mask2d.shape == (early.root_num, early._A):
            roots = early.reset_mask(mask2d, keep_inference=True)
        else:
            roots = early.reset(legal_actions, keep_inference=True)
        if bool(_g(self._cfg, "collect_with_pure_policy", False)):
            # efficientzero.py:597,644-655: no search; sample from softmax(policy logits over the legal actions)
            if early is None:
                network_output = self._collect_model.initial_inference(data, roots)
                pred_values, logits = np.asarray(network_output.value, np.float32), np.asarray(network_output.policy_logits)
            else:
                pred_values = np.zeros(active_collect_env_num, np.float32)
                logits = np.zeros((active_collect_env_num, self._collect_model.action_space_size), np.float32)
                L.check(L.lib().lz_roots_get_root_outputs(roots._h, pred_values, logits.reshape(-1)))
            pred_values = pred_values.reshape(active_collect_env_num, 1)
            policy_logits = logits.tolist()
            for i, env_id in enumerate(ready_env_id):
                z = np.asarray([policy_logits[i][a] for a in legal_actions[i]], np.float32)
                e = np.exp(z - z.max())
                policy_values = (e / e.sum()).tolist()
                policy_values = policy_values / np.sum(policy_values)
                idx = L.rs().choice(len(legal_actions[i]), p=policy_values)
                action = np.where(np.asarray(action_mask[i]) == 1.0)[0][idx]
                output[env_id] = {'action': action, 'searched_value': pred_values[i], 'predicted_value': pred_values[i],
                                  'predicted_policy_logits': policy_logits[i]}
            return output
        alpha = self._mcfg["root_dirichlet_alpha"]
        fused = getattr(self._collect_model, "_is_lz_engine_model", False) and hasattr(roots, "get_search_results")
        counts = [len(l) for l in legal_actions]
        if fused and self._device_noise:
            noises = None
        elif len(set(counts)) == 1:  # one vectorised draw instead of one np.random.dirichlet call per env (efficientzero.py:599-602)
            noises = L.rs().dirichlet([alpha] * counts[0], size=active_collect_env_num).astype(np.float32)
        else:
            noises = [L.rs().dirichlet([alpha] * c).astype(np.float32) for c in counts]
        if fused:
            # no read-back (and no synchronisation) before the search: predictions come back with the search results
            if early is None:
                self._collect_model.initial_inference(data, roots, fetch=False)
            if noises is None:
                roots.prepare_from_inference_dirichlet(self._mcfg["root_noise_weight"], alpha, to_play)
            else:
                roots.prepare_from_inference(self._mcfg["root_noise_weight"], noises, to_play)
            self._search(self._mcts_collect, roots, self._collect_model, _HbmTokens(roots), to_play)
            eps_cfg0 = _g(self._cfg, "eps", {}) or {}
            if self._device_select:  # select_action on the device, inside the same read-back
                dist, cnt, roots_values, pred_values, logits, dev_pos, dev_ent = roots.get_search_results(
                    select=(self._collect_mcts_temperature, bool(_g(eps_cfg0, "eps_greedy_exploration_in_collect", False))))
            else:
                dist, cnt, roots_values, pred_values, logits = roots.get_search_results()
            roots_visit_count_distributions = [r[:c] for r, c in zip(dist.tolist(), cnt.tolist())]
            policy_logits = logits.tolist()
        else:
            network_output = self._collect_model.initial_inference(data, roots)
            pred_values, policy_logits = network_output.value, network_output.policy_logits.tolist()
            roots.prepare_from_inference(self._mcfg["root_noise_weight"], noises, to_play)
            self._search(self._mcts_collect, roots, self._collect_model, network_output, to_play)
            roots_visit_count_distributions = roots.get_distributions()
            roots_values = roots.get_values()
        pred_values = list(np.asarray(pred_values, np.float32).reshape(active_collect_env_num, 1))  # efficientzero.py:583: [B, 1]
        eps_cfg = _g(self._cfg, "eps", {}) or {}
        eps_greedy = bool(_g(eps_cfg, "eps_greedy_exploration_in_collect", False))
        if self._device_select and not fused:
            dev_pos, dev_ent = roots.select_action(self._collect_mcts_temperature, deterministic=eps_greedy)
        if self._device_select:
            dev_pos, dev_ent = np.asarray(dev_pos).tolist(), np.asarray(dev_ent, np.float64).tolist()
        roots_values = list(roots_values)
        for i, env_id in enumerate(ready_env_id):
            distributions, value = roots_visit_count_distributions[i], roots_values[i]
            if self._device_select:
                idx, entropy = dev_pos[i], dev_ent[i]
                action = legal_actions[i][idx]
                if eps_greedy and L.rs().rand() < self.collect_epsilon:
                    action = L.rs().choice(legal_actions[i])
            elif eps_greedy:
                idx, entropy = select_action(distributions, temperature=self._collect_mcts_temperature, deterministic=True)
                action = np.where(np.asarray(action_mask[i]) == 1.0)[0][idx]
                if L.rs().rand() < self.collect_epsilon:
                    action = L.rs().choice(legal_actions[i])
            else:
                idx, entropy = select_action(distributions, temperature=self._collect_mcts_temperature, deterministic=False)
                action = np.where(np.asarray(action_mask[i]) == 1.0)[0][idx]
            output[env_id] = {
                'action': action,
                'visit_count_distributions': distributions,
                'visit_count_distribution_entropy': entropy,
                'searched_value': value,
                'predicted_value': pred_values[i],
                'predicted_policy_logits': policy_logits[i],
            }
        return output

    def forward_collect_rows(self, data, action_mask, rows_out, temperature=1, to_play=[-1], timestep=None, frame_floats=None,
                             epsilon=0.25):
        """``forward_collect_rows_begin`` + ``forward_collect_rows_end`` (see there)"""
        return self.forward_collect_rows_end(self.forward_collect_rows_begin(data, action_mask, rows_out, temperature=temperature, to_play=to_play,
                                                                             timestep=timestep, frame_floats=frame_floats, epsilon=epsilon))

    def forward_collect_rows_begin(self, data, action_mask, rows_out, temperature=1, to_play=[-1], timestep=None, frame_floats=None,
                                   epsilon=0.25):
        """The collect forward for a vectorised collector: same search as ``_forward_collect`` (engine model, device
        tensors), but what comes back is not a dict per env: ``rows_out`` -- a float32 [B, W] tensor IN HBM (W =
        shard.row_width(A, frame_floats)) -- receives the packed env-step rows (action, search statistics, action mask, to_play,
        newest observation frame: the GameSegment field set) straight from the device, and the return value is the [B, 8 + 2A]
        header block on the host: ``header[:, shard.F_ACTION]`` steps the environments, ``GameSegmentBatch.store_search_stats_rows
        (header)`` does the per-step bookkeeping of muzero_collector.py:588-620 for all envs at once.  No per-env Python loop:
        one np.nonzero, one Dirichlet draw, one read-back.

        Two halves: ``_begin`` ENQUEUES the whole forward -- representation network, noise, prepare, the 50 simulations,
        select_action + row packing -- and returns a ticket without waiting for the device; ``_end(ticket)`` waits for those rows (their
        event, not the stream) and returns the header.  Between the two the caller does host work (segment bookkeeping of the previous
        step) or enqueues ANOTHER env group's forward behind this one, so the device never waits for the host
        (lightzero_amd.worker.MuZeroVectorCollector)."""
        from .. import shard
        self.collect_epsilon = epsilon
        model = self._collect_model
        B, A = data.shape[0], model.action_space_size
        mask = np.asarray(action_mask)
        if bool(_g(self._cfg, "collect_with_pure_policy", False)):
            return dict(done=self._pure_policy_rows(data, mask, rows_out, to_play, timestep, frame_floats))
        roots = self._roots_cache.get(B)
        if roots is None:
            roots = self._roots(B, [np.nonzero(mask[j])[0].tolist() for j in range(B)])
            model.initial_inference(data, roots, fetch=False)
        else:
            model.initial_inference(data, roots, fetch=False)   # launched before the host touches the mask
            roots.reset_mask(mask, keep_inference=True)
        counts = (mask != 0).sum(1)
        alpha = self._mcfg["root_dirichlet_alpha"]
        tp = list(to_play) if len(to_play) == B else [to_play[0]] * B
        if self._device_noise:
            roots.prepare_from_inference_dirichlet(self._mcfg["root_noise_weight"], alpha, tp)
        else:
            if (counts == counts[0]).all():
                noises = L.rs().dirichlet([alpha] * int(counts[0]), size=B).astype(np.float32)
            else:  # ragged: one gamma draw for the whole batch, normalised per root (what np.random.dirichlet does per env)
                g = L.rs().gamma(alpha, size=int(counts.sum()))
                seg = np.repeat(np.arange(B), counts)
                noises = (g / np.bincount(seg, weights=g, minlength=B)[seg]).astype(np.float32)
            roots.prepare_from_inference(self._mcfg["root_noise_weight"], noises, tp)
        self._search(self._mcts_collect, roots, model, _HbmTokens(roots), tp)
        if frame_floats is None:
            frame_floats = rows_out.shape[1] - shard.HEADER - 2 * A
        eps_cfg = _g(self._cfg, "eps", {}) or {}
        eps_greedy = bool(_g(eps_cfg, "eps_greedy_exploration_in_collect", False))
        # the observation pointer is passed explicitly: the library's cached pointer of the last initial inference would outlive a
        # tensor the caller's allocator has recycled in between
        roots.collect_rows_begin(temperature, eps_greedy, rows_out.data_ptr(), rows_out.shape[1], frame_floats, timestep=timestep,
                                 d_obs_ptr=data.data_ptr() if hasattr(data, "data_ptr") and getattr(data, "is_cuda", False) else None)
        return dict(roots=roots, rows_out=rows_out, mask=mask, eps_greedy=eps_greedy, epsilon=epsilon, B=B, data=data)   # (data: kept alive while the device reads it)

    def forward_collect_rows_end(self, ticket):
        from .. import shard
        if "done" in ticket:
            return ticket["done"]
        roots, rows_out, mask, eps_greedy, epsilon, B = (ticket[k] for k in ("roots", "rows_out", "mask", "eps_greedy", "epsilon", "B"))
        header, _ = roots.collect_rows_end()
        if eps_greedy:
            # efficientzero.py:622-632: arg-max of the visit counts, replaced by a uniformly random LEGAL action with probability
            # collect_epsilon -- for all envs at once; the action word of the device rows is patched too
            explore = np.nonzero(L.rs().rand(B) < float(epsilon))[0]
            if explore.size:
                legal = mask[explore] != 0
                pick = (L.rs().rand(explore.size) * legal.sum(1)).astype(np.int64)             # position in the legal list
                acts = np.argmax(np.cumsum(legal, 1) > pick[:, None], 1).astype(np.float32)       # -> action index
                header[explore, shard.F_ACTION] = acts
                import torch
                rows_out[torch.as_tensor(explore, device=rows_out.device), shard.F_ACTION] = torch.as_tensor(acts, device=rows_out.device)
        return header

    def _pure_policy_rows(self, data, mask, rows_out, to_play, timestep, frame_floats):
        """``collect_with_pure_policy`` on the rows path (efficientzero.py:597,644-655; muzero_collector.py:98-99,505,596): no search -- the
        action is drawn from softmax(policy logits over the legal actions), the stored search statistics are the collector's
        ``temp_visit_list`` (zeros) and the predicted value (``store_search_stats(temp_visit_list, pred_value)``).  One inference, one
        vectorised draw for all envs, the rows assembled on the device (header upload + the newest frame sliced from ``data``)."""
        import torch
        from .. import shard
        model = self._collect_model
        B, A = data.shape[0], model.action_space_size
        roots = self._roots_cache.get(B)
        if roots is None:
            roots = self._roots(B, [np.nonzero(mask[j])[0].tolist() for j in range(B)])
        model.initial_inference(data, roots, fetch=False)
        pred = np.zeros(B, np.float32)
        logits = np.zeros((B, A), np.float32)
        L.check(L.lib().lz_roots_get_root_outputs(roots._h, pred, logits.reshape(-1)))
        legal = mask != 0
        # torch.softmax over the legal entries (float32: max-subtracted exponentials), renormalised in float64 like the reference's
        # ``policy_values / np.sum(policy_values)``; one inverse-CDF draw per env instead of np.random.choice per env
        z = np.where(legal, logits, -np.inf).astype(np.float32)
        e = np.exp(z - z.max(1, keepdims=True)).astype(np.float32)
        p = (e / e.sum(1, keepdims=True)).astype(np.float64)
        p /= p.sum(1, keepdims=True)
        u = L.rs().rand(B)
        actions = np.minimum((np.cumsum(p, 1) < u[:, None]).sum(1), A - 1)
        # a draw can only land on a legal action (zero-probability entries add nothing to the CDF); guard the float edge u ~ 1
        last_legal = A - 1 - np.argmax(legal[:, ::-1], 1)
        actions = np.where(legal[np.arange(B), actions], actions, last_legal)
        tp = np.asarray(list(to_play) if len(to_play) == B else [to_play[0]] * B, np.float32)
        header = np.zeros((B, shard.HEADER + 2 * A), np.float32)
        header[:, shard.F_ACTION] = actions
        header[:, shard.F_ROOT_VALUE] = pred      # 'searched_value': pred_values (efficientzero.py:651)
        header[:, shard.F_PRED_VALUE] = pred
        header[:, shard.F_TO_PLAY] = tp
        header[:, shard.F_TIMESTEP] = -1 if timestep is None else np.asarray(timestep, np.float32)
        header[:, shard.F_N_LEGAL] = legal.sum(1)
        header[:, shard.HEADER + A:shard.HEADER + 2 * A] = mask   # child visits stay zero (temp_visit_list), entropy 0
        if rows_out is not None:
            if frame_floats is None:
                frame_floats = rows_out.shape[1] - shard.HEADER - 2 * A
            HW = shard.HEADER + 2 * A
            rows_out[:, :HW] = torch.from_numpy(header).to(rows_out.device, non_blocking=True)
            if frame_floats > 0 and hasattr(data, "data_ptr"):
                rows_out[:, HW:HW + frame_floats] = data.reshape(B, -1)[:, -frame_floats:].to(rows_out.device)
        return header

    def _forward_eval(self, data, action_mask, to_play=[-1], ready_env_id=None, **kwargs):
        active_eval_env_num = data.shape[0]
        if ready_env_id is None:
            ready_env_id = np.arange(active_eval_env_num)
        output = {i: None for i in ready_env_id}
        to_play = list(to_play) if len(to_play) == active_eval_env_num else [to_play[0]] * active_eval_env_num
        legal_actions = [np.nonzero(action_mask[j])[0].tolist() for j in range(active_eval_env_num)]
        roots = self._roots(active_eval_env_num, legal_actions)
        fused = getattr(self._eval_model, "_is_lz_engine_model", False) and hasattr(roots, "get_search_results")
        if fused:  # like the collect forward: nothing is read back before the search, one synchronisation after it
            self._eval_model.initial_inference(data, roots, fetch=False)
            roots.prepare_from_inference_no_noise(to_play)  # efficientzero.py:721
            self._search(self._mcts_eval, roots, self._eval_model, _HbmTokens(roots), to_play)
            res = roots.get_search_results(select=(1, True)) if self._device_select else roots.get_search_results()
            dist, cnt, roots_values, pred_values, logits = res[:5]
            if self._device_select:
                dev_pos, dev_ent = res[5], res[6]
            roots_visit_count_distributions = [dist[i, :cnt[i]].tolist() for i in range(active_eval_env_num)]
            policy_logits = logits.tolist()
        else:
            network_output = self._eval_model.initial_inference(data, roots)
            pred_values, policy_logits = network_output.value, network_output.policy_logits.tolist()
            roots.prepare_from_inference_no_noise(to_play)  # efficientzero.py:721
            self._search(self._mcts_eval, roots, self._eval_model, network_output, to_play)
            roots_visit_count_distributions = roots.get_distributions()
            roots_values = roots.get_values()
            if self._device_select:
                dev_pos, dev_ent = roots.select_action(1, deterministic=True)
        pred_values = np.asarray(pred_values, np.float32).reshape(active_eval_env_num, 1)  # efficientzero.py:711: [B, 1]
        for i, env_id in enumerate(ready_env_id):
            distributions, value = roots_visit_count_distributions[i], roots_values[i]
            if self._device_select:
                idx, entropy = int(dev_pos[i]), float(dev_ent[i])
            else:
                idx, entropy = select_action(distributions, temperature=1, deterministic=True)  # efficientzero.py:733
            action = np.where(np.asarray(action_mask[i]) == 1.0)[0][idx]
            output[env_id] = {
                'action': action,
                'visit_count_distributions': distributions,
                'visit_count_distribution_entropy': entropy,
                'searched_value': value,
                'predicted_value': pred_values[i],
                'predicted_policy_logits': policy_logits[i],
            }
        return output
