"""TEST INFRASTRUCTURE ONLY -- restatement of the reference's Python drivers on the hot path, used as
the end-to-end checker and as the timed CPU baseline (bench.py ``cpu_baseline``):

* ``ez_search``          EfficientZeroMCTSCtree.search   lzero/mcts/tree_search/mcts_ctree.py:745-876
* ``ez_forward_collect`` EfficientZeroPolicy._forward_collect up to get_distributions/get_values
                         lzero/policy/efficientzero.py:572-615

``tree`` is any module with the reference's Cython surface: the compiled reference itself
(oracle/build_ref.load) or the C restatement (oracle/ctree.ez_tree).  ``model`` is the torch fp32
restatement (oracle/torch_models.EfficientZeroModel).  The per-simulation host<->device round trips,
list conversions and the per-root Python gather loop of the reference are kept, because they are what
the reference spends its time on.
"""
import numpy as np
import torch

from .torch_models import InverseScalarTransform


def ez_search(tree, roots, model, latent_state_roots, reward_hidden_state_roots, to_play_batch, cfg, device="cpu",
              record=None, ist=None):
    """cfg: dict(num_simulations, pb_c_base, pb_c_init, discount_factor, value_delta_max, lstm_horizon_len).
    ``ist``: replaces the value_inverse_scalar_transform_handle (tests replay recorded post-transform scalars through it)."""
    ist = ist or InverseScalarTransform(cfg.get("support_range", (-300., 301., 1.)), cfg.get("categorical_distribution", True), device=device)  # the policy builds it from cfg.model.*_support_range
    with torch.no_grad():
        model.eval()
        batch_size = roots.num
        pb_c_base, pb_c_init, discount_factor = cfg["pb_c_base"], cfg["pb_c_init"], cfg["discount_factor"]
        latent_state_batch_in_search_path = [latent_state_roots]
        reward_hidden_state_c_batch = [reward_hidden_state_roots[0]]
        reward_hidden_state_h_batch = [reward_hidden_state_roots[1]]
        min_max_stats_lst = tree.MinMaxStatsList(batch_size)
        min_max_stats_lst.set_delta(cfg["value_delta_max"])
        for simulation_index in range(cfg["num_simulations"]):
            latent_states, hidden_states_c_reward, hidden_states_h_reward = [], [], []
            results = tree.ResultsWrapper(batch_size)
            ix_l, iy_l, last_actions, virtual_to_play_batch = tree.batch_traverse(
                roots, pb_c_base, pb_c_init, discount_factor, min_max_stats_lst, results, list(to_play_batch))
            search_lens = results.get_search_len()
            for ix, iy in zip(ix_l, iy_l):  # mcts_ctree.py:815-818
                latent_states.append(latent_state_batch_in_search_path[ix][iy])
                hidden_states_c_reward.append(reward_hidden_state_c_batch[ix][0][iy])
                hidden_states_h_reward.append(reward_hidden_state_h_batch[ix][0][iy])
            latent_states = torch.from_numpy(np.asarray(latent_states)).to(device)
            hidden_states_c_reward = torch.from_numpy(np.asarray(hidden_states_c_reward)).to(device).unsqueeze(0)
            hidden_states_h_reward = torch.from_numpy(np.asarray(hidden_states_h_reward)).to(device).unsqueeze(0)
            last_actions_t = torch.from_numpy(np.asarray(last_actions)).to(device).long()
            out = model.recurrent_inference(latent_states, (hidden_states_c_reward, hidden_states_h_reward), last_actions_t)
            latent = out.latent_state.detach().cpu().numpy()
            policy_logits = out.policy_logits.detach().cpu().numpy()
            value = ist(out.value).detach().cpu().numpy()
            value_prefix = ist(out.value_prefix).detach().cpu().numpy()
            rhs = (out.reward_hidden_state[0].detach().cpu().numpy(), out.reward_hidden_state[1].detach().cpu().numpy())
            latent_state_batch_in_search_path.append(latent)
            value_prefix_batch = value_prefix.reshape(-1).tolist()
            value_batch = value.reshape(-1).tolist()
            policy_logits_batch = policy_logits.tolist()
            reset_idx = (np.array(search_lens) % cfg["lstm_horizon_len"] == 0)  # mcts_ctree.py:859
            rhs[0][:, reset_idx, :] = 0
            rhs[1][:, reset_idx, :] = 0
            is_reset_list = reset_idx.astype(np.int32).tolist()
            reward_hidden_state_c_batch.append(rhs[0])
            reward_hidden_state_h_batch.append(rhs[1])
            if record is not None:
                record.append(dict(ix=list(ix_l), action=list(last_actions), search_len=list(search_lens),
                                   value_prefix=value_prefix.reshape(-1).copy(), value=value.reshape(-1).copy(),
                                   policy_logits=policy_logits.copy()))
            tree.batch_backpropagate(simulation_index + 1, discount_factor, value_prefix_batch, value_batch,
                                     policy_logits_batch, min_max_stats_lst, results, is_reset_list, virtual_to_play_batch)


def ez_forward_collect(tree, model, obs, legal_actions, noises, to_play, cfg, device="cpu", roots_kwargs=None,
                       record=None):
    """obs: torch [B,C,H,W] on ``device``.  Returns (visit-count distributions, root values, predicted values,
    policy logits) like efficientzero.py:582-615."""
    ist = InverseScalarTransform(cfg.get("support_range", (-300., 301., 1.)), cfg.get("categorical_distribution", True), device=device)  # the policy builds it from cfg.model.*_support_range
    with torch.no_grad():
        model.eval()
        out = model.initial_inference(obs)
        pred_values = ist(out.value).detach().cpu().numpy()
        latent_state_roots = out.latent_state.detach().cpu().numpy()
        reward_hidden_state_roots = (out.reward_hidden_state[0].detach().cpu().numpy(),
                                     out.reward_hidden_state[1].detach().cpu().numpy())
        policy_logits = out.policy_logits.detach().cpu().numpy().tolist()
        roots = tree.Roots(obs.shape[0], legal_actions, **(roots_kwargs or {}))
        if noises is not None:
            roots.prepare(cfg["root_noise_weight"], noises, list(out.value_prefix), policy_logits, list(to_play))
        else:
            roots.prepare_no_noise(list(out.value_prefix), policy_logits, list(to_play))
        ez_search(tree, roots, model, latent_state_roots, reward_hidden_state_roots, to_play, cfg, device, record)
        return roots.get_distributions(), roots.get_values(), pred_values.reshape(-1), policy_logits


def mz_search(tree, roots, model, latent_state_roots, to_play_batch, cfg, device="cpu", deterministic=None, ist=None, record=None):
    """MuZeroMCTSCtree.search  lzero/mcts/tree_search/mcts_ctree.py:267-368 (the reference calls
    recurrent_inference twice per simulation, :338 and :340-345, and discards the first result; it is called once
    here, which leaves the outputs unchanged)."""
    ist = ist or InverseScalarTransform(cfg.get("support_range", (-300., 301., 1.)), cfg.get("categorical_distribution", True), device=device)  # the policy builds it from cfg.model.*_support_range
    with torch.no_grad():
        model.eval()
        batch_size = roots.num
        pb_c_base, pb_c_init, discount_factor = cfg["pb_c_base"], cfg["pb_c_init"], cfg["discount_factor"]
        latent_state_batch_in_search_path = [latent_state_roots]
        min_max_stats_lst = tree.MinMaxStatsList(batch_size)
        min_max_stats_lst.set_delta(cfg["value_delta_max"])
        kw = {} if deterministic is None else dict(deterministic=deterministic)
        for simulation_index in range(cfg["num_simulations"]):
            latent_states = []
            results = tree.ResultsWrapper(batch_size)
            ix_l, iy_l, last_actions, virtual_to_play_batch = tree.batch_traverse(
                roots, pb_c_base, pb_c_init, discount_factor, min_max_stats_lst, results, list(to_play_batch), **kw)
            for ix, iy in zip(ix_l, iy_l):
                latent_states.append(latent_state_batch_in_search_path[ix][iy])
            latent_states = torch.from_numpy(np.asarray(latent_states)).to(device)
            last_actions_t = torch.from_numpy(np.asarray(last_actions)).to(device).long()
            out = model.recurrent_inference(latent_states, last_actions_t)
            latent_state_batch_in_search_path.append(out.latent_state.detach().cpu().numpy())
            value = ist(out.value).detach().cpu().numpy()
            reward = ist(out.reward).detach().cpu().numpy()
            if record is not None:   # (the reward under the EfficientZero records' key: tests/e2e_attrib.py reads both families)
                record.append(dict(ix=list(ix_l), action=list(last_actions), search_len=list(results.get_search_len()),
                                   value_prefix=reward.reshape(-1).copy(), value=value.reshape(-1).copy(),
                                   policy_logits=out.policy_logits.detach().cpu().numpy().copy()))
            tree.batch_backpropagate(simulation_index + 1, discount_factor, reward.reshape(-1).tolist(),
                                     value.reshape(-1).tolist(), out.policy_logits.detach().cpu().numpy().tolist(),
                                     min_max_stats_lst, results, virtual_to_play_batch)


def mz_forward_collect(tree, model, obs, legal_actions, noises, to_play, cfg, device="cpu", roots_kwargs=None,
                       deterministic=None, record=None):
    """MuZeroPolicy._forward_collect up to get_distributions/get_values  lzero/policy/muzero.py:745-790."""
    ist = InverseScalarTransform(cfg.get("support_range", (-300., 301., 1.)), cfg.get("categorical_distribution", True), device=device)  # the policy builds it from cfg.model.*_support_range
    with torch.no_grad():
        model.eval()
        out = model.initial_inference(obs)
        pred_values = ist(out.value).detach().cpu().numpy()
        latent_state_roots = out.latent_state.detach().cpu().numpy()
        policy_logits = out.policy_logits.detach().cpu().numpy().tolist()
        roots = tree.Roots(obs.shape[0], legal_actions, **(roots_kwargs or {}))
        if noises is not None:
            roots.prepare(cfg["root_noise_weight"], noises, list(out.reward), policy_logits, list(to_play))
        else:
            roots.prepare_no_noise(list(out.reward), policy_logits, list(to_play))
        mz_search(tree, roots, model, latent_state_roots, to_play, cfg, device, deterministic, record=record)
        return roots.get_distributions(), roots.get_values(), pred_values.reshape(-1), policy_logits


def sez_search(tree, roots, model, latent_state_roots, reward_hidden_state_roots, to_play_batch, cfg, device="cpu"):
    """SampledEfficientZeroMCTSCtree.search (continuous)  lzero/mcts/tree_search/mcts_ctree_sampled.py:480-600."""
    ist = InverseScalarTransform(device=device)
    with torch.no_grad():
        model.eval()
        batch_size = roots.num
        pb_c_base, pb_c_init, discount_factor = cfg["pb_c_base"], cfg["pb_c_init"], cfg["discount_factor"]
        latent_state_batch_in_search_path = [latent_state_roots]
        reward_hidden_state_c_pool = [reward_hidden_state_roots[0]]
        reward_hidden_state_h_pool = [reward_hidden_state_roots[1]]
        min_max_stats_lst = tree.MinMaxStatsList(batch_size)
        min_max_stats_lst.set_delta(cfg["value_delta_max"])
        for simulation_index in range(cfg["num_simulations"]):
            latent_states, hidden_states_c_reward, hidden_states_h_reward = [], [], []
            results = tree.ResultsWrapper(batch_size)
            ix_l, iy_l, last_actions, virtual_to_play_batch = tree.batch_traverse(
                roots, pb_c_base, pb_c_init, discount_factor, min_max_stats_lst, results, list(to_play_batch), True)
            search_lens = results.get_search_len()
            for ix, iy in zip(ix_l, iy_l):
                latent_states.append(latent_state_batch_in_search_path[ix][iy])
                hidden_states_c_reward.append(reward_hidden_state_c_pool[ix][0][iy])
                hidden_states_h_reward.append(reward_hidden_state_h_pool[ix][0][iy])
            latent_states = torch.from_numpy(np.asarray(latent_states)).to(device)
            hidden_states_c_reward = torch.from_numpy(np.asarray(hidden_states_c_reward)).to(device).unsqueeze(0)
            hidden_states_h_reward = torch.from_numpy(np.asarray(hidden_states_h_reward)).to(device).unsqueeze(0)
            last_actions_t = torch.from_numpy(np.asarray(last_actions, np.float32)).to(device)
            out = model.recurrent_inference(latent_states, (hidden_states_c_reward, hidden_states_h_reward), last_actions_t)
            latent_state_batch_in_search_path.append(out.latent_state.detach().cpu().numpy())
            value = ist(out.value).detach().cpu().numpy()
            value_prefix = ist(out.value_prefix).detach().cpu().numpy()
            rhs = (out.reward_hidden_state[0].detach().cpu().numpy(), out.reward_hidden_state[1].detach().cpu().numpy())
            reset_idx = (np.array(search_lens) % cfg["lstm_horizon_len"] == 0)
            rhs[0][:, reset_idx, :] = 0
            rhs[1][:, reset_idx, :] = 0
            reward_hidden_state_c_pool.append(rhs[0])
            reward_hidden_state_h_pool.append(rhs[1])
            tree.batch_backpropagate(simulation_index + 1, discount_factor, value_prefix.reshape(-1).tolist(),
                                     value.reshape(-1).tolist(), out.policy_logits.detach().cpu().numpy().tolist(),
                                     min_max_stats_lst, results, reset_idx.astype(np.int32).tolist(), virtual_to_play_batch)


def ez_search_with_reuse(tree, roots, model, latent_state_roots, reward_hidden_state_roots, to_play_batch, cfg, true_action_list,
                         reuse_value_list, device="cpu"):
    """EfficientZeroMCTSCtree.search_with_reuse  lzero/mcts/tree_search/mcts_ctree.py:878-1002 (ReZero).  One deviation:
    is_reset_list is computed per ROOT from its own search length; the reference builds it from the packed search lengths
    (:968-971) and cnode.cpp:645 indexes it by root, which reads past its end whenever a root skips inference."""
    ist = InverseScalarTransform(device=device)
    with torch.no_grad():
        model.eval()
        batch_size = roots.num
        pb_c_base, pb_c_init, discount_factor = cfg["pb_c_base"], cfg["pb_c_init"], cfg["discount_factor"]
        latent_state_batch_in_search_path = [latent_state_roots]
        reward_hidden_state_c_batch = [reward_hidden_state_roots[0]]
        reward_hidden_state_h_batch = [reward_hidden_state_roots[1]]
        min_max_stats_lst = tree.MinMaxStatsList(batch_size)
        min_max_stats_lst.set_delta(cfg["value_delta_max"])
        infer_sum = 0
        for simulation_index in range(cfg["num_simulations"]):
            latent_states, hidden_states_c_reward, hidden_states_h_reward = [], [], []
            temp_actions, temp_search_lens, no_inference_lst, reuse_lst = [], [], [], []
            results = tree.ResultsWrapper(batch_size)
            ix_l, iy_l, last_actions, virtual_to_play_batch = tree.batch_traverse_with_reuse(
                roots, pb_c_base, pb_c_init, discount_factor, min_max_stats_lst, results, list(to_play_batch),
                list(true_action_list), list(reuse_value_list))
            search_lens = results.get_search_len()
            for count, (ix, iy) in enumerate(zip(ix_l, iy_l)):
                if ix != -1:
                    latent_states.append(latent_state_batch_in_search_path[ix][iy])
                    hidden_states_c_reward.append(reward_hidden_state_c_batch[ix][0][iy])
                    hidden_states_h_reward.append(reward_hidden_state_h_batch[ix][0][iy])
                    temp_actions.append(last_actions[count])
                    temp_search_lens.append(search_lens[count])
                else:
                    no_inference_lst.append(iy)
                if ix == 0 and last_actions[count] == true_action_list[count]:
                    reuse_lst.append(count)
            length = len(temp_actions)
            if length != 0:
                out = model.recurrent_inference(
                    torch.from_numpy(np.asarray(latent_states)).to(device),
                    (torch.from_numpy(np.asarray(hidden_states_c_reward)).to(device).unsqueeze(0),
                     torch.from_numpy(np.asarray(hidden_states_h_reward)).to(device).unsqueeze(0)),
                    torch.from_numpy(np.asarray(temp_actions)).to(device).long())
                latent_state_batch_in_search_path.append(out.latent_state.detach().cpu().numpy())
                value_batch = ist(out.value).detach().cpu().numpy().reshape(-1).tolist()
                value_prefix_batch = ist(out.value_prefix).detach().cpu().numpy().reshape(-1).tolist()
                policy_logits_batch = out.policy_logits.detach().cpu().numpy().tolist()
                rhs = (out.reward_hidden_state[0].detach().cpu().numpy(), out.reward_hidden_state[1].detach().cpu().numpy())
                reset_idx = (np.array(temp_search_lens) % cfg["lstm_horizon_len"] == 0)
                rhs[0][:, reset_idx, :] = 0
                rhs[1][:, reset_idx, :] = 0
                reward_hidden_state_c_batch.append(rhs[0])
                reward_hidden_state_h_batch.append(rhs[1])
            else:
                latent_state_batch_in_search_path.append([])
                value_batch, policy_logits_batch, value_prefix_batch = [], [], []
                reward_hidden_state_c_batch.append([])
                reward_hidden_state_h_batch.append([])
            is_reset_list = (np.array(search_lens) % cfg["lstm_horizon_len"] == 0).astype(np.int32).tolist()
            no_inference_lst.append(-1)
            reuse_lst.append(-1)
            tree.batch_backpropagate_with_reuse(simulation_index + 1, discount_factor, value_prefix_batch, value_batch,
                                                policy_logits_batch, min_max_stats_lst, results, is_reset_list,
                                                virtual_to_play_batch, no_inference_lst, reuse_lst, list(reuse_value_list))
            infer_sum += length
        return length, infer_sum / cfg["num_simulations"]


def gmz_search(tree, roots, model, latent_state_roots, to_play_batch, cfg, device="cpu"):
    """GumbelMuZeroMCTSCtree.search  lzero/mcts/tree_search/mcts_ctree.py:1067-1172."""
    ist = InverseScalarTransform(device=device)
    with torch.no_grad():
        model.eval()
        batch_size = roots.num
        discount_factor = cfg["discount_factor"]
        latent_state_batch_in_search_path = [latent_state_roots]
        min_max_stats_lst = tree.MinMaxStatsList(batch_size)
        min_max_stats_lst.set_delta(cfg["value_delta_max"])
        for simulation_index in range(cfg["num_simulations"]):
            latent_states = []
            results = tree.ResultsWrapper(batch_size)
            ix_l, iy_l, last_actions, virtual_to_play_batch = tree.batch_traverse(
                roots, cfg["num_simulations"], cfg["max_num_considered_actions"], discount_factor, results, list(to_play_batch))
            for ix, iy in zip(ix_l, iy_l):
                latent_states.append(latent_state_batch_in_search_path[ix][iy])
            latent_states = torch.from_numpy(np.asarray(latent_states)).to(device)
            last_actions_t = torch.from_numpy(np.asarray(last_actions)).to(device).long()
            out = model.recurrent_inference(latent_states, last_actions_t)
            latent_state_batch_in_search_path.append(out.latent_state.detach().cpu().numpy())
            value = ist(out.value).detach().cpu().numpy()
            reward = ist(out.reward).detach().cpu().numpy()
            tree.batch_back_propagate(simulation_index + 1, discount_factor, reward.reshape(-1).tolist(), value.reshape(-1).tolist(),
                                      out.policy_logits.detach().cpu().numpy().tolist(), min_max_stats_lst, results,
                                      virtual_to_play_batch)
