"""TEST INFRASTRUCTURE: EfficientZeroPolicy._forward_collect / _forward_eval of the reference (lzero/policy/efficientzero.py:539-657,
670-747) with ONLY the module's import lines changed -- the two `from lzero...` imports of the names these bodies use
(`EfficientZeroMCTSCtree as MCTSCtree`; `select_action`, `ez_network_output_unpack`) point at lightzero_amd.  The bodies are the
reference's statements, docstrings and comments dropped (generated with ast.unparse; tests/test_reference_forward_cpu.py checks
them against /root/reference statement by statement where it exists).  tests/test_reference_forward_gpu.py runs them on an engine
model: INTEGRATION.md section 1 -- the reference's collect forward runs unmodified on the drop-in."""
from typing import List, Union

import numpy as np
import torch

from lightzero_amd.mcts.tree_search.mcts_ctree import EfficientZeroMCTSCtree as MCTSCtree   # reference: from lzero.mcts import EfficientZeroMCTSCtree as MCTSCtree
from lightzero_amd.policy.utils import select_action, ez_network_output_unpack              # reference: from lzero.policy import ... select_action, ez_network_output_unpack ...


class ReferenceForwardBodies(object):
    """holder of the two methods; a test builds the attributes they read (``_cfg``, ``_collect_model`` / ``_eval_model``,
    ``_mcts_collect`` / ``_mcts_eval``, ``value_inverse_scalar_transform_handle``) the way _init_collect / _init_eval do"""

    def _forward_collect(self, data: torch.Tensor, action_mask: list=None, temperature: float=1, to_play: List=[-1], epsilon: float=0.25, ready_env_id: np.array=None, **kwargs):
        self._collect_model.eval()
        self._collect_mcts_temperature = temperature
        self.collect_epsilon = epsilon
        active_collect_env_num = data.shape[0]
        if ready_env_id is None:
            ready_env_id = np.arange(active_collect_env_num)
        output = {i: None for i in ready_env_id}
        with torch.no_grad():
            network_output = self._collect_model.initial_inference(data)
            (latent_state_roots, value_prefix_roots, reward_hidden_state_roots, pred_values, policy_logits) = ez_network_output_unpack(network_output)
            pred_values = self.value_inverse_scalar_transform_handle(pred_values).detach().cpu().numpy()
            latent_state_roots = latent_state_roots.detach().cpu().numpy()
            reward_hidden_state_roots = (reward_hidden_state_roots[0].detach().cpu().numpy(), reward_hidden_state_roots[1].detach().cpu().numpy())
            policy_logits = policy_logits.detach().cpu().numpy().tolist()
            legal_actions = [np.nonzero(action_mask[j])[0].tolist() for j in range(active_collect_env_num)]
            if not self._cfg.collect_with_pure_policy:
                noises = [np.random.dirichlet([self._cfg.root_dirichlet_alpha] * int(sum(action_mask[j]))).astype(np.float32).tolist() for j in range(active_collect_env_num)]
                if self._cfg.mcts_ctree:
                    roots = MCTSCtree.roots(active_collect_env_num, legal_actions)
                else:
                    roots = MCTSPtree.roots(active_collect_env_num, legal_actions)
                roots.prepare(self._cfg.root_noise_weight, noises, value_prefix_roots, policy_logits, to_play)
                self._mcts_collect.search(roots, self._collect_model, latent_state_roots, reward_hidden_state_roots, to_play)
                roots_visit_count_distributions = roots.get_distributions()
                roots_values = roots.get_values()
                for (i, env_id) in enumerate(ready_env_id):
                    (distributions, value) = (roots_visit_count_distributions[i], roots_values[i])
                    if self._cfg.eps.eps_greedy_exploration_in_collect:
                        (action_index_in_legal_action_set, visit_count_distribution_entropy) = select_action(distributions, temperature=self._collect_mcts_temperature, deterministic=True)
                        action = np.where(action_mask[i] == 1.0)[0][action_index_in_legal_action_set]
                        if np.random.rand() < self.collect_epsilon:
                            action = np.random.choice(legal_actions[i])
                    else:
                        (action_index_in_legal_action_set, visit_count_distribution_entropy) = select_action(distributions, temperature=self._collect_mcts_temperature, deterministic=False)
                        action = np.where(action_mask[i] == 1.0)[0][action_index_in_legal_action_set]
                    output[env_id] = {'action': action, 'visit_count_distributions': distributions, 'visit_count_distribution_entropy': visit_count_distribution_entropy, 'searched_value': value, 'predicted_value': pred_values[i], 'predicted_policy_logits': policy_logits[i]}
            else:
                for (i, env_id) in enumerate(ready_env_id):
                    policy_values = torch.softmax(torch.tensor([policy_logits[i][a] for a in legal_actions[i]]), dim=0).tolist()
                    policy_values = policy_values / np.sum(policy_values)
                    action_index_in_legal_action_set = np.random.choice(len(legal_actions[i]), p=policy_values)
                    action = np.where(action_mask[i] == 1.0)[0][action_index_in_legal_action_set]
                    output[env_id] = {'action': action, 'searched_value': pred_values[i], 'predicted_value': pred_values[i], 'predicted_policy_logits': policy_logits[i]}
        return output

    def _forward_eval(self, data: torch.Tensor, action_mask: list, to_play: Union[int, List]=[-1], ready_env_id: np.array=None, **kwargs):
        self._eval_model.eval()
        active_eval_env_num = data.shape[0]
        if ready_env_id is None:
            ready_env_id = np.arange(active_eval_env_num)
        output = {i: None for i in ready_env_id}
        with torch.no_grad():
            network_output = self._eval_model.initial_inference(data)
            (latent_state_roots, value_prefix_roots, reward_hidden_state_roots, pred_values, policy_logits) = ez_network_output_unpack(network_output)
            if not self._eval_model.training:
                pred_values = self.value_inverse_scalar_transform_handle(pred_values).detach().cpu().numpy()
                latent_state_roots = latent_state_roots.detach().cpu().numpy()
                reward_hidden_state_roots = (reward_hidden_state_roots[0].detach().cpu().numpy(), reward_hidden_state_roots[1].detach().cpu().numpy())
                policy_logits = policy_logits.detach().cpu().numpy().tolist()
            legal_actions = [np.nonzero(action_mask[j])[0].tolist() for j in range(active_eval_env_num)]
            if self._cfg.mcts_ctree:
                roots = MCTSCtree.roots(active_eval_env_num, legal_actions)
            else:
                roots = MCTSPtree.roots(active_eval_env_num, legal_actions)
            roots.prepare_no_noise(value_prefix_roots, policy_logits, to_play)
            self._mcts_eval.search(roots, self._eval_model, latent_state_roots, reward_hidden_state_roots, to_play)
            roots_visit_count_distributions = roots.get_distributions()
            roots_values = roots.get_values()
            for (i, env_id) in enumerate(ready_env_id):
                (distributions, value) = (roots_visit_count_distributions[i], roots_values[i])
                (action_index_in_legal_action_set, visit_count_distribution_entropy) = select_action(distributions, temperature=1, deterministic=True)
                action = np.where(action_mask[i] == 1.0)[0][action_index_in_legal_action_set]
                output[env_id] = {'action': action, 'visit_count_distributions': distributions, 'visit_count_distribution_entropy': visit_count_distribution_entropy, 'searched_value': value, 'predicted_value': pred_values[i], 'predicted_policy_logits': policy_logits[i]}
        return output
