"""Drop-in for the EfficientZero/MuZero drivers of lzero/mcts/tree_search/mcts_ctree.py.

``EfficientZeroMCTSCtree(cfg).roots(n, legal_actions)`` / ``.search(roots, model, latent_state_roots,
reward_hidden_state_roots, to_play_batch)`` keep the reference's names, arguments and in-place effect
on ``roots``.  Two execution paths:

* ``model`` is an engine model (lightzero_amd.model.EfficientZeroModel) and the roots were filled by
  its ``initial_inference``: the whole ``num_simulations`` loop runs on the device (lz_search) --
  latent/LSTM pools, gathers, recurrent inference, h^-1, LSTM reset, expand and backup never leave HBM.
* any other ``model`` exposing ``recurrent_inference`` (e.g. a torch module): the reference's loop
  (mcts_ctree.py:782-876) is executed with the HBM-resident tree kernels doing batch_traverse /
  batch_backpropagate -- the plumbing configuration.
"""
import copy

import numpy as np

from ..ctree.ctree_efficientzero import ez_tree as tree_efficientzero
from ..ctree.ctree_muzero import mz_tree as tree_muzero
from ... import _lib as L


class _Cfg(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def _get(cfg, name, default=None):
    if isinstance(cfg, dict):
        return cfg.get(name, default)
    return getattr(cfg, name, default)


def _inverse_scalar_transform(logits, support_min, epsilon=0.001, categorical=True, step=1.0):
    """InverseScalarTransform.__call__ (lzero/policy/scaling_transform.py:82-92) with the same torch ops, for the
    foreign-model path (``logits``: torch tensor [B, support], or [B, 1] scalars when not categorical); returns numpy [B, 1]."""
    import torch
    if categorical:
        support = (support_min + step * torch.arange(logits.shape[1], dtype=torch.float32, device=logits.device)).unsqueeze(0)
        value_probs = torch.softmax(logits, dim=1)
        value = value_probs.mul_(support).sum(1, keepdim=True)
    else:
        value = logits
    tmp = ((torch.sqrt(1 + 4 * epsilon * (torch.abs(value) + 1 + epsilon)) - 1) / (2 * epsilon))
    return (torch.sign(value) * (tmp * tmp - 1)).detach().cpu().numpy()


def _np(x):
    """a handle may return a torch tensor (the reference's InverseScalarTransform does) or an array"""
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


class _InverseScalarTransformHandle(object):
    """what the reference drivers keep as ``value_inverse_scalar_transform_handle`` / ``reward_inverse_scalar_transform_handle``
    (mcts_ctree.py:726-729: one per support, built from cfg.model.{value,reward}_support_range and categorical_distribution)"""

    def __init__(self, support_range, categorical_distribution=True):
        self.support_min, self.step = float(support_range[0]), float(support_range[2]) if len(support_range) > 2 else 1.0
        self.categorical_distribution = bool(categorical_distribution)

    def __call__(self, logits):
        if getattr(logits, "_lz_inverse_transformed", False):
            # the root value of an engine model's initial_inference(obs): h^-1 was applied on the device.  The reference's call
            # sites continue with ``.detach().cpu().numpy()`` (efficientzero.py:587), so a tensor goes back
            return logits
        return _inverse_scalar_transform(logits, self.support_min, categorical=self.categorical_distribution, step=self.step)


def _make_handles(self):
    model_cfg = _get(self._cfg, "model", {}) or {}
    cat = bool(_get(model_cfg, "categorical_distribution", True))
    vrange = _get(model_cfg, "value_support_range", (-300., 301., 1.))
    rrange = _get(model_cfg, "reward_support_range", vrange)
    self._support_min = float(vrange[0])
    self._categorical = cat
    self.value_inverse_scalar_transform_handle = _InverseScalarTransformHandle(vrange, cat)
    self.reward_inverse_scalar_transform_handle = _InverseScalarTransformHandle(rrange, cat)


def _fused(roots, model, latent_state_roots, num_simulations, what):
    """True: ``model`` is an engine model and the root state is in HBM -> the fused device loop.  Either ``roots`` were filled by
    the model's ``initial_inference(obs, roots)``, or ``latent_state_roots`` is the token of ``model.initial_inference(obs)`` (the
    reference's call order) and the inference is adopted into the prepared roots here.
    False: a foreign model -- anything with ``recurrent_inference`` taking arrays, an engine model's Python method included -> the
    reference loop around the device tree.  An engine model paired with roots / tokens that are not its own is a caller error."""
    token = latent_state_roots if getattr(latent_state_roots, "_is_lz_hbm_token", False) else None
    if not getattr(model, "_is_lz_engine_model", False):
        if token is not None:
            raise L.LzError("%s: latent_state_roots is an HBM token of an engine model, but `model` is %s" % (what, type(model).__name__))
        return False
    if token is not None:
        if token.model is not model:
            raise L.LzError("%s: the latent-state token belongs to another model object than the one searching" % what)
        roots._adopt(token.roots, model, num_simulations)
        return True
    if getattr(roots, "_inferred_by", None) is model:
        return True
    if getattr(roots, "_inferred_by", None) is not None:
        raise L.LzError("%s: these roots were inferred by another model object; an engine model searches the roots its own "
                        "initial_inference filled" % what)
    if isinstance(latent_state_roots, tuple) and len(latent_state_roots) == 2 and latent_state_roots[0] == "hbm-pool":
        raise L.LzError("%s: the latent-state token refers to other roots than the ones given (model.initial_inference(obs, roots) "
                        "fills exactly the roots it is handed)" % what)
    return False   # arrays: the engine model is driven through its Python recurrent_inference like any other model


class EfficientZeroMCTSCtree(object):
    config = dict(root_dirichlet_alpha=0.3, root_noise_weight=0.25, pb_c_base=19652, pb_c_init=1.25,
                  value_delta_max=0.01)

    @classmethod
    def default_config(cls):
        cfg = _Cfg(copy.deepcopy(cls.config))
        cfg["cfg_type"] = cls.__name__ + "Dict"
        return cfg

    def __init__(self, cfg=None):
        default_config = self.default_config()
        if cfg is not None:
            default_config.update(dict(cfg) if isinstance(cfg, dict) else {k: getattr(cfg, k) for k in dir(cfg)
                                                                            if not k.startswith("_")})
        self._cfg = default_config
        _make_handles(self)

    @classmethod
    def roots(cls, active_collect_env_num, legal_actions, action_space_size=None, max_simulations=None):
        return tree_efficientzero.Roots(active_collect_env_num, legal_actions, action_space_size=action_space_size,
                                        max_simulations=max_simulations)

    def search(self, roots, model, latent_state_roots, reward_hidden_state_roots, to_play_batch):
        cfg = self._cfg
        num_simulations = int(cfg["num_simulations"])
        if _fused(roots, model, latent_state_roots, num_simulations, "EfficientZeroMCTSCtree.search"):
            L.check(L.lib().lz_search(roots._h, num_simulations, int(cfg["pb_c_base"]), float(cfg["pb_c_init"]),
                                      float(cfg["discount_factor"]), int(cfg["lstm_horizon_len"]),
                                      float(cfg["value_delta_max"])))
            return
        self._search_foreign_model(roots, model, latent_state_roots, reward_hidden_state_roots, to_play_batch)

    # the reference's loop with a foreign model; tree on the device
    def _search_foreign_model(self, roots, model, latent_state_roots, reward_hidden_state_roots, to_play_batch):
        import torch
        cfg = self._cfg
        device = _get(cfg, "device", "cpu")
        with torch.no_grad():
            model.eval()
            batch_size = roots.num
            pb_c_base, pb_c_init, discount_factor = cfg["pb_c_base"], cfg["pb_c_init"], cfg["discount_factor"]
            latent_pool = [np.asarray(latent_state_roots)]
            c_pool = [np.asarray(reward_hidden_state_roots[0])]
            h_pool = [np.asarray(reward_hidden_state_roots[1])]
            min_max_stats_lst = tree_efficientzero.MinMaxStatsList(batch_size)
            min_max_stats_lst.set_delta(cfg["value_delta_max"])
            ar = np.arange(batch_size)
            for simulation_index in range(int(cfg["num_simulations"])):
                results = tree_efficientzero.ResultsWrapper(num=batch_size)
                tp = to_play_batch if _get(cfg, "env_type", "not_board_games") == "not_board_games" else copy.deepcopy(to_play_batch)
                ix, iy, last_actions, virtual_to_play_batch = tree_efficientzero.batch_traverse(
                    roots, pb_c_base, pb_c_init, discount_factor, min_max_stats_lst, results, tp)
                search_lens = results.get_search_len()
                ix = np.asarray(ix); iy = np.asarray(iy)
                lat_all = np.stack(latent_pool); c_all = np.stack(c_pool); h_all = np.stack(h_pool)
                latent_states = torch.from_numpy(lat_all[ix, iy]).to(device)
                hc = torch.from_numpy(c_all[ix, 0, iy]).to(device).unsqueeze(0)
                hh = torch.from_numpy(h_all[ix, 0, iy]).to(device).unsqueeze(0)
                out = model.recurrent_inference(latent_states, (hc, hh), torch.from_numpy(np.asarray(last_actions)).to(device).long())
                latent_pool.append(out.latent_state.detach().cpu().numpy())
                value = _np(self.value_inverse_scalar_transform_handle(out.value))
                value_prefix = _np(self.value_inverse_scalar_transform_handle(out.value_prefix))
                rhs = [out.reward_hidden_state[0].detach().cpu().numpy().copy(), out.reward_hidden_state[1].detach().cpu().numpy().copy()]
                reset_idx = (np.array(search_lens) % int(cfg["lstm_horizon_len"]) == 0)
                rhs[0][:, reset_idx, :] = 0
                rhs[1][:, reset_idx, :] = 0
                c_pool.append(rhs[0]); h_pool.append(rhs[1])
                tree_efficientzero.batch_backpropagate(
                    simulation_index + 1, discount_factor, value_prefix.reshape(-1), value.reshape(-1),
                    out.policy_logits.detach().cpu().numpy(), min_max_stats_lst, results,
                    reset_idx.astype(np.int32), virtual_to_play_batch)


    def search_with_reuse(self, roots, model, latent_state_roots, reward_hidden_state_roots, to_play_batch,
                          true_action_list=None, reuse_value_list=None):
        """EfficientZeroMCTSCtree.search_with_reuse (mcts_ctree.py:878-1002, ReZero): returns the reference's
        ``(length, average_infer)``.  Engine model: the whole loop on the device (lz_search_with_reuse).  Foreign model:
        the reference loop with the device tree; ``is_reset`` is computed per root from its own search length (the
        reference indexes a packed list by root there, which reads out of bounds whenever a root skips inference)."""
        cfg = self._cfg
        S = int(cfg["num_simulations"])
        if _fused(roots, model, latent_state_roots, S, "EfficientZeroMCTSCtree.search_with_reuse"):
            import ctypes
            length, avg = ctypes.c_int(0), ctypes.c_double(0.0)
            L.check(L.lib().lz_search_with_reuse(roots._h, S, int(cfg["pb_c_base"]), float(cfg["pb_c_init"]),
                                                 float(cfg["discount_factor"]), int(cfg["lstm_horizon_len"]),
                                                 float(cfg["value_delta_max"]), L.i32(true_action_list), L.f32(reuse_value_list),
                                                 ctypes.byref(length), ctypes.byref(avg)))
            return length.value, avg.value
        import torch
        T = tree_efficientzero
        device = _get(cfg, "device", "cpu")
        with torch.no_grad():
            model.eval()
            batch_size = roots.num
            pb_c_base, pb_c_init, discount_factor = cfg["pb_c_base"], cfg["pb_c_init"], cfg["discount_factor"]
            latent_pool = [np.asarray(latent_state_roots)]
            c_pool = [np.asarray(reward_hidden_state_roots[0])]
            h_pool = [np.asarray(reward_hidden_state_roots[1])]
            mm = T.MinMaxStatsList(batch_size)
            mm.set_delta(cfg["value_delta_max"])
            infer_sum, length = 0, 0
            for simulation_index in range(S):
                results = T.ResultsWrapper(num=batch_size)
                tp = to_play_batch if _get(cfg, "env_type", "not_board_games") == "not_board_games" else copy.deepcopy(to_play_batch)
                ix, iy, last_actions, vtp = T.batch_traverse_with_reuse(roots, pb_c_base, pb_c_init, discount_factor, mm, results, tp,
                                                                        true_action_list, reuse_value_list)
                search_lens = results.get_search_len()
                need = [k for k in range(batch_size) if ix[k] != -1]
                no_inference_lst = [iy[k] for k in range(batch_size) if ix[k] == -1] + [-1]
                reuse_lst = [k for k in range(batch_size) if ix[k] == 0 and last_actions[k] == true_action_list[k]] + [-1]
                length = len(need)
                if length:
                    lat = np.stack([latent_pool[ix[k]][iy[k]] for k in need])
                    hc = np.stack([c_pool[ix[k]][0][iy[k]] for k in need])
                    hh = np.stack([h_pool[ix[k]][0][iy[k]] for k in need])
                    out = model.recurrent_inference(torch.from_numpy(lat).to(device),
                                                    (torch.from_numpy(hc).to(device).unsqueeze(0), torch.from_numpy(hh).to(device).unsqueeze(0)),
                                                    torch.from_numpy(np.asarray([last_actions[k] for k in need])).to(device).long())
                    latent_pool.append(out.latent_state.detach().cpu().numpy())
                    value = _np(self.value_inverse_scalar_transform_handle(out.value)).reshape(-1)
                    value_prefix = _np(self.value_inverse_scalar_transform_handle(out.value_prefix)).reshape(-1)
                    policy = out.policy_logits.detach().cpu().numpy()
                    rhs = [out.reward_hidden_state[0].detach().cpu().numpy().copy(), out.reward_hidden_state[1].detach().cpu().numpy().copy()]
                    reset_packed = (np.array([search_lens[k] for k in need]) % int(cfg["lstm_horizon_len"]) == 0)
                    rhs[0][:, reset_packed, :] = 0
                    rhs[1][:, reset_packed, :] = 0
                    c_pool.append(rhs[0]); h_pool.append(rhs[1])
                else:
                    latent_pool.append([]); c_pool.append([]); h_pool.append([])
                    value, value_prefix, policy = [], [], []
                is_reset_list = (np.array(search_lens) % int(cfg["lstm_horizon_len"]) == 0).astype(np.int32)
                T.batch_backpropagate_with_reuse(simulation_index + 1, discount_factor, value_prefix, value, policy, mm, results,
                                                 is_reset_list, vtp, no_inference_lst, reuse_lst, reuse_value_list)
                infer_sum += length
        return length, infer_sum / S


class MuZeroMCTSCtree(object):
    """lzero/mcts/tree_search/mcts_ctree.py:211-368 (reference loop; tree kernels on the device)."""
    config = dict(root_dirichlet_alpha=0.3, root_noise_weight=0.25, pb_c_base=19652, pb_c_init=1.25,
                  value_delta_max=0.01)

    @classmethod
    def default_config(cls):
        cfg = _Cfg(copy.deepcopy(cls.config))
        cfg["cfg_type"] = cls.__name__ + "Dict"
        return cfg

    def __init__(self, cfg=None):
        default_config = self.default_config()
        if cfg is not None:
            default_config.update(dict(cfg))
        self._cfg = default_config
        _make_handles(self)

    @classmethod
    def roots(cls, active_collect_env_num, legal_actions, action_space_size=None, max_simulations=None):
        return tree_muzero.Roots(active_collect_env_num, legal_actions, action_space_size=action_space_size,
                                 max_simulations=max_simulations)

    def search(self, roots, model, latent_state_roots, to_play_batch, task_id=None):
        cfg = self._cfg
        if _fused(roots, model, latent_state_roots, int(cfg["num_simulations"]), "MuZeroMCTSCtree.search"):
            # fused path: every simulation on the device (the reference runs recurrent_inference twice per
            # simulation, mcts_ctree.py:338-345; once is enough)
            L.check(L.lib().lz_search(roots._h, int(cfg["num_simulations"]), int(cfg["pb_c_base"]), float(cfg["pb_c_init"]),
                                      float(cfg["discount_factor"]), 0, float(cfg["value_delta_max"])))
            return
        import torch
        device = _get(cfg, "device", "cpu")
        with torch.no_grad():
            model.eval()
            batch_size = roots.num
            pb_c_base, pb_c_init, discount_factor = cfg["pb_c_base"], cfg["pb_c_init"], cfg["discount_factor"]
            latent_pool = [np.asarray(latent_state_roots)]
            min_max_stats_lst = tree_muzero.MinMaxStatsList(batch_size)
            min_max_stats_lst.set_delta(cfg["value_delta_max"])
            for simulation_index in range(int(cfg["num_simulations"])):
                results = tree_muzero.ResultsWrapper(num=batch_size)
                tp = to_play_batch if _get(cfg, "env_type", "not_board_games") == "not_board_games" else copy.deepcopy(to_play_batch)
                ix, iy, last_actions, virtual_to_play_batch = tree_muzero.batch_traverse(
                    roots, pb_c_base, pb_c_init, discount_factor, min_max_stats_lst, results, tp)
                lat_all = np.stack(latent_pool)
                latent_states = torch.from_numpy(lat_all[np.asarray(ix), np.asarray(iy)]).to(device)
                out = model.recurrent_inference(latent_states, torch.from_numpy(np.asarray(last_actions)).to(device).long())
                latent_pool.append(out.latent_state.detach().cpu().numpy())
                # the handles apply h^-1 in both modes (categorical_distribution=False: to the scalar itself, scaling_transform.py:85-91)
                value = _np(self.value_inverse_scalar_transform_handle(out.value))
                reward = _np(self.reward_inverse_scalar_transform_handle(out.reward))
                tree_muzero.batch_backpropagate(simulation_index + 1, discount_factor, reward.reshape(-1), value.reshape(-1),
                                                out.policy_logits.detach().cpu().numpy(), min_max_stats_lst, results,
                                                virtual_to_play_batch)

    def search_with_reuse(self, roots, model, latent_state_roots, to_play_batch, true_action_list=None, reuse_value_list=None):
        """MuZeroMCTSCtree.search_with_reuse (mcts_ctree.py:370-470, ReZero): returns ``(length, average_infer)``."""
        cfg = self._cfg
        S = int(cfg["num_simulations"])
        if _fused(roots, model, latent_state_roots, S, "MuZeroMCTSCtree.search_with_reuse"):
            import ctypes
            length, avg = ctypes.c_int(0), ctypes.c_double(0.0)
            L.check(L.lib().lz_search_with_reuse(roots._h, S, int(cfg["pb_c_base"]), float(cfg["pb_c_init"]),
                                                 float(cfg["discount_factor"]), 0, float(cfg["value_delta_max"]),
                                                 L.i32(true_action_list), L.f32(reuse_value_list), ctypes.byref(length), ctypes.byref(avg)))
            return length.value, avg.value
        import torch
        T = tree_muzero
        device = _get(cfg, "device", "cpu")
        with torch.no_grad():
            model.eval()
            batch_size = roots.num
            pb_c_base, pb_c_init, discount_factor = cfg["pb_c_base"], cfg["pb_c_init"], cfg["discount_factor"]
            latent_pool = [np.asarray(latent_state_roots)]
            mm = T.MinMaxStatsList(batch_size)
            mm.set_delta(cfg["value_delta_max"])
            infer_sum, length = 0, 0
            for simulation_index in range(S):
                results = T.ResultsWrapper(num=batch_size)
                tp = to_play_batch if _get(cfg, "env_type", "not_board_games") == "not_board_games" else copy.deepcopy(to_play_batch)
                ix, iy, last_actions, vtp = T.batch_traverse_with_reuse(roots, pb_c_base, pb_c_init, discount_factor, mm, results, tp,
                                                                        true_action_list, reuse_value_list)
                need = [k for k in range(batch_size) if ix[k] != -1]
                no_inference_lst = [iy[k] for k in range(batch_size) if ix[k] == -1] + [-1]
                reuse_lst = [k for k in range(batch_size) if ix[k] == 0 and last_actions[k] == true_action_list[k]] + [-1]
                length = len(need)
                if length:
                    lat = np.stack([latent_pool[ix[k]][iy[k]] for k in need])
                    out = model.recurrent_inference(torch.from_numpy(lat).to(device),
                                                    torch.from_numpy(np.asarray([last_actions[k] for k in need])).to(device).long())
                    latent_pool.append(out.latent_state.detach().cpu().numpy())
                    value = _np(self.value_inverse_scalar_transform_handle(out.value)).reshape(-1)
                    reward = _np(self.reward_inverse_scalar_transform_handle(out.reward)).reshape(-1)
                    policy = out.policy_logits.detach().cpu().numpy()
                else:
                    latent_pool.append([])
                    value, reward, policy = [], [], []
                T.batch_backpropagate_with_reuse(simulation_index + 1, discount_factor, reward, value, policy, mm, results, vtp,
                                                 no_inference_lst, reuse_lst, reuse_value_list)
                infer_sum += length
        return length, infer_sum / S


class SampledEfficientZeroMCTSCtree(object):
    """lzero/mcts/tree_search/mcts_ctree_sampled.py:394+ (continuous action spaces): the reference loop with the
    HBM-resident Sampled-EfficientZero tree kernels doing batch_traverse / batch_backpropagate.  ``model`` is any
    module with the SampledEfficientZeroModelMLP inference contract (``recurrent_inference(latent, (h, c), action)``
    returning value / value_prefix logits, ``policy_logits`` = (mu | sigma), latent_state, reward_hidden_state)."""
    config = dict(root_dirichlet_alpha=0.3, root_noise_weight=0.25, pb_c_base=19652, pb_c_init=1.25,
                  value_delta_max=0.01)

    @classmethod
    def default_config(cls):
        cfg = _Cfg(copy.deepcopy(cls.config))
        cfg["cfg_type"] = cls.__name__ + "Dict"
        return cfg

    def __init__(self, cfg=None):
        default_config = self.default_config()
        if cfg is not None:
            default_config.update(dict(cfg))
        self._cfg = default_config
        _make_handles(self)

    @classmethod
    def roots(cls, active_collect_env_num, legal_actions, action_space_size, num_of_sampled_actions,
              continuous_action_space=True, max_simulations=None):
        from ..ctree.ctree_sampled_efficientzero import ezs_tree
        return ezs_tree.Roots(active_collect_env_num, legal_actions, action_space_size, num_of_sampled_actions,
                              continuous_action_space, max_simulations=max_simulations)

    def search(self, roots, model, latent_state_roots, reward_hidden_state_roots, to_play_batch):
        import torch
        from ..ctree.ctree_sampled_efficientzero import ezs_tree
        cfg = self._cfg
        if _fused(roots, model, latent_state_roots, int(cfg["num_simulations"]), "SampledEfficientZeroMCTSCtree.search"):
            # engine model: the whole loop (select, MLP + LSTM inference, sampling of the leaf's K actions, backup) on the device
            L.check(L.lib().lz_search(roots._h, int(cfg["num_simulations"]), int(cfg["pb_c_base"]), float(cfg["pb_c_init"]),
                                      float(cfg["discount_factor"]), int(cfg["lstm_horizon_len"]), float(cfg["value_delta_max"])))
            return
        device = _get(cfg, "device", "cpu")
        with torch.no_grad():
            model.eval()
            batch_size = roots.num
            pb_c_base, pb_c_init, discount_factor = cfg["pb_c_base"], cfg["pb_c_init"], cfg["discount_factor"]
            latent_pool = [np.asarray(latent_state_roots)]
            c_pool = [np.asarray(reward_hidden_state_roots[0])]
            h_pool = [np.asarray(reward_hidden_state_roots[1])]
            min_max_stats_lst = ezs_tree.MinMaxStatsList(batch_size)
            min_max_stats_lst.set_delta(cfg["value_delta_max"])
            for simulation_index in range(int(cfg["num_simulations"])):
                results = ezs_tree.ResultsWrapper(num=batch_size)
                tp = to_play_batch if _get(cfg, "env_type", "not_board_games") == "not_board_games" else copy.deepcopy(to_play_batch)
                ix, iy, last_actions, virtual_to_play_batch = ezs_tree.batch_traverse(
                    roots, pb_c_base, pb_c_init, discount_factor, min_max_stats_lst, results, tp, roots.continuous)
                search_lens = results.get_search_len()
                ix = np.asarray(ix); iy = np.asarray(iy)
                latent_states = torch.from_numpy(np.stack(latent_pool)[ix, iy]).to(device)
                hc = torch.from_numpy(np.stack(c_pool)[ix, 0, iy]).to(device).unsqueeze(0)
                hh = torch.from_numpy(np.stack(h_pool)[ix, 0, iy]).to(device).unsqueeze(0)
                la = torch.from_numpy(np.asarray(last_actions, np.float32)).to(device)
                if not roots.continuous:
                    la = la.long()  # mcts_ctree_sampled.py: discrete actions are fed as indices
                out = model.recurrent_inference(latent_states, (hc, hh), la)
                latent_pool.append(out.latent_state.detach().cpu().numpy())
                value = _np(self.value_inverse_scalar_transform_handle(out.value))
                value_prefix = _np(self.value_inverse_scalar_transform_handle(out.value_prefix))
                rhs = [out.reward_hidden_state[0].detach().cpu().numpy().copy(), out.reward_hidden_state[1].detach().cpu().numpy().copy()]
                reset_idx = (np.array(search_lens) % int(cfg["lstm_horizon_len"]) == 0)
                rhs[0][:, reset_idx, :] = 0
                rhs[1][:, reset_idx, :] = 0
                c_pool.append(rhs[0]); h_pool.append(rhs[1])
                ezs_tree.batch_backpropagate(simulation_index + 1, discount_factor, value_prefix.reshape(-1), value.reshape(-1),
                                             out.policy_logits.detach().cpu().numpy(), min_max_stats_lst, results,
                                             reset_idx.astype(np.int32), virtual_to_play_batch)


class GumbelMuZeroMCTSCtree(object):
    """lzero/mcts/tree_search/mcts_ctree.py:1004-1172 (Gumbel MuZero): engine MuZero model -> lz_gsearch, whole loop on the device;
    any other model -> the reference loop with the device tree (gmz_tree drop-in)."""
    config = dict(root_dirichlet_alpha=0.3, root_noise_weight=0.25, pb_c_base=19652, pb_c_init=1.25, value_delta_max=0.01,
                  max_num_considered_actions=4)

    @classmethod
    def default_config(cls):
        cfg = _Cfg(copy.deepcopy(cls.config))
        cfg["cfg_type"] = cls.__name__ + "Dict"
        return cfg

    def __init__(self, cfg=None):
        default_config = self.default_config()
        if cfg is not None:
            default_config.update(dict(cfg))
        self._cfg = default_config
        _make_handles(self)

    @classmethod
    def roots(cls, active_collect_env_num, legal_actions, action_space_size=None, max_simulations=None):
        from ..ctree.ctree_gumbel_muzero import gmz_tree
        return gmz_tree.Roots(active_collect_env_num, legal_actions, action_space_size=action_space_size, max_simulations=max_simulations)

    def search(self, roots, model, latent_state_roots, to_play_batch):
        from ..ctree.ctree_gumbel_muzero import gmz_tree
        cfg = self._cfg
        S, m, discount_factor = int(cfg["num_simulations"]), int(cfg["max_num_considered_actions"]), cfg["discount_factor"]
        if _fused(roots, model, latent_state_roots, S, "GumbelMuZeroMCTSCtree.search"):
            L.check(L.lib().lz_gsearch(roots._h, S, m, float(discount_factor)))
            return
        import torch
        device = _get(cfg, "device", "cpu")
        with torch.no_grad():
            model.eval()
            batch_size = roots.num
            latent_pool = [np.asarray(latent_state_roots)]
            mm = gmz_tree.MinMaxStatsList(batch_size)
            mm.set_delta(cfg["value_delta_max"])
            for simulation_index in range(S):
                results = gmz_tree.ResultsWrapper(num=batch_size)
                tp = to_play_batch if _get(cfg, "env_type", "not_board_games") == "not_board_games" else copy.deepcopy(to_play_batch)
                ix, iy, last_actions, vtp = gmz_tree.batch_traverse(roots, S, m, discount_factor, results, tp)
                lat = np.stack(latent_pool)[np.asarray(ix), np.asarray(iy)]
                out = model.recurrent_inference(torch.from_numpy(lat).to(device), torch.from_numpy(np.asarray(last_actions)).to(device).long())
                latent_pool.append(out.latent_state.detach().cpu().numpy())
                value = _np(self.value_inverse_scalar_transform_handle(out.value)).reshape(-1)
                reward = _np(self.reward_inverse_scalar_transform_handle(out.reward)).reshape(-1)
                gmz_tree.batch_back_propagate(simulation_index + 1, discount_factor, reward, value,
                                              out.policy_logits.detach().cpu().numpy(), mm, results, vtp)
