"""Collect half of lzero/policy/sampled_efficientzero.py::SampledEfficientZeroPolicy (``_forward_collect`` :797-935,
continuous action spaces) with the HBM-resident Sampled-EfficientZero tree; ``model`` is any module with the
SampledEfficientZeroModelMLP inference contract (torch).  Same arguments and per-env output dict as the reference
(incl. ``root_sampled_actions``)."""
import numpy as np

from ..mcts.tree_search.mcts_ctree import SampledEfficientZeroMCTSCtree as MCTSCtree, _inverse_scalar_transform
from .efficientzero import _g
from .utils import select_action


class SampledEfficientZeroPolicy(object):
    def __init__(self, cfg, model):
        self._cfg = cfg
        self._collect_model = model
        mc = _g(cfg, "model", {}) or {}
        self._A = int(_g(mc, "action_space_size"))
        self._K = int(_g(mc, "num_of_sampled_actions", 20))
        self._support_min = float(_g(mc, "value_support_range", (-300., 301., 1.))[0])
        self._mcfg = dict(num_simulations=_g(cfg, "num_simulations", 50), discount_factor=_g(cfg, "discount_factor", 0.997),
                          lstm_horizon_len=_g(cfg, "lstm_horizon_len", 5), pb_c_base=_g(cfg, "pb_c_base", 19652),
                          pb_c_init=_g(cfg, "pb_c_init", 1.25), value_delta_max=_g(cfg, "value_delta_max", 0.01),
                          root_dirichlet_alpha=_g(cfg, "root_dirichlet_alpha", 0.3),
                          root_noise_weight=_g(cfg, "root_noise_weight", 0.25),
                          env_type=_g(cfg, "env_type", "not_board_games"), device=_g(cfg, "device", "cpu"), model=mc)
        self._mcts_collect = MCTSCtree(self._mcfg)
        self._collect_mcts_temperature = 1.

    def _forward_collect(self, data, action_mask=None, temperature=1, to_play=[-1], epsilon=0.25, ready_env_id=None,
                         **kwargs):
        import torch
        self._collect_mcts_temperature = temperature
        n = data.shape[0]
        if ready_env_id is None:
            ready_env_id = np.arange(n)
        output = {i: None for i in ready_env_id}
        to_play = list(to_play) if len(to_play) == n else [to_play[0]] * n
        with torch.no_grad():
            self._collect_model.eval()
            out = self._collect_model.initial_inference(data)
            pred_values = _inverse_scalar_transform(out.value, self._support_min)
            latent_state_roots = out.latent_state.detach().cpu().numpy()
            reward_hidden_state_roots = (out.reward_hidden_state[0].detach().cpu().numpy(),
                                         out.reward_hidden_state[1].detach().cpu().numpy())
            policy_logits = out.policy_logits.detach().cpu().numpy().tolist()
        legal_actions = [[-1 for _ in range(self._K)] for _ in range(n)]
        roots = MCTSCtree.roots(n, legal_actions, self._A, self._K, True, max_simulations=int(self._mcfg["num_simulations"]))
        noises = [np.random.dirichlet([self._mcfg["root_dirichlet_alpha"]] * self._K).astype(np.float32).tolist() for _ in range(n)]
        roots.prepare(self._mcfg["root_noise_weight"], noises, list(out.value_prefix), policy_logits, to_play)
        self._mcts_collect.search(roots, self._collect_model, latent_state_roots, reward_hidden_state_roots, to_play)
        roots_visit_count_distributions = roots.get_distributions()
        roots_values = roots.get_values()
        roots_sampled_actions = roots.get_sampled_actions()
        for i, env_id in enumerate(ready_env_id):
            distributions, value = roots_visit_count_distributions[i], roots_values[i]
            root_sampled_actions = np.array([a for a in roots_sampled_actions[i]])
            idx, entropy = select_action(distributions, temperature=self._collect_mcts_temperature, deterministic=False)
            output[env_id] = {
                'action': np.array(roots_sampled_actions[i][idx]),
                'visit_count_distributions': distributions,
                'root_sampled_actions': root_sampled_actions,
                'visit_count_distribution_entropy': entropy,
                'searched_value': value,
                'predicted_value': pred_values[i],
                'predicted_policy_logits': policy_logits[i],
            }
        return output
