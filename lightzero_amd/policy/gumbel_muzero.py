"""Collect / eval halves of lzero/policy/gumbel_muzero.py::GumbelMuZeroPolicy (``_forward_collect`` :500-607, ``_forward_eval``
:620-710) on the MI355X engine: MuZero model + Gumbel MuZero tree.  Same arguments and per-env output dict as the reference
(``roots_completed_value`` and ``improved_policy_probs`` included); the action is the arg-max of the improved policy."""
import numpy as np

from .. import _lib as L
from ..mcts.tree_search.mcts_ctree import GumbelMuZeroMCTSCtree as MCTSCtree
from .efficientzero import _g, _mcts_seed
from .utils import CheckpointIngest, select_action


class GumbelMuZeroPolicy(CheckpointIngest):
    def __init__(self, cfg, model):
        self._cfg = cfg
        self._collect_model = model
        self._eval_model = model
        self._mcfg = dict(num_simulations=_g(cfg, "num_simulations", 50), discount_factor=_g(cfg, "discount_factor", 0.997),
                          max_num_considered_actions=_g(cfg, "max_num_considered_actions", 4),
                          value_delta_max=_g(cfg, "value_delta_max", 0.01), root_dirichlet_alpha=_g(cfg, "root_dirichlet_alpha", 0.3),
                          root_noise_weight=_g(cfg, "root_noise_weight", 0.25), env_type=_g(cfg, "env_type", "not_board_games"),
                          model=_g(cfg, "model", {}) or {})
        self._mcts_collect = MCTSCtree(self._mcfg)
        self._mcts_eval = MCTSCtree(self._mcfg)
        self._collect_mcts_temperature = 1.
        self._roots_cache = {}

    def forward(self, *args, **kwargs):
        return self._forward_collect(*args, **kwargs)

    def _roots(self, n, legal_actions):
        roots = self._roots_cache.get(n)
        if roots is None:
            roots = MCTSCtree.roots(n, legal_actions, action_space_size=self._collect_model.action_space_size,
                                    max_simulations=int(self._mcfg["num_simulations"]))
            self._roots_cache[n] = roots
        else:
            roots.reset(legal_actions)
        return roots

    def _run(self, mcts, model, data, action_mask, to_play, noise, temperature, deterministic, ready_env_id):
        n = data.shape[0]
        if ready_env_id is None:
            ready_env_id = np.arange(n)
        to_play = list(to_play) if len(to_play) == n else [to_play[0]] * n
        legal_actions = [np.nonzero(action_mask[j])[0].tolist() for j in range(n)]
        roots = self._roots(n, legal_actions)
        out = model.initial_inference(data, roots)
        pred_values, policy_logits = out.value, out.policy_logits.tolist()
        if noise:
            alpha = self._mcfg["root_dirichlet_alpha"]
            noises = [L.rs().dirichlet([alpha] * int(sum(action_mask[j]))).astype(np.float32).tolist() for j in range(n)]
            roots.prepare_from_inference(self._mcfg["root_noise_weight"], noises, to_play)
        else:
            roots.prepare_from_inference_no_noise(to_play)
        mcts.search(roots, model, out.latent_state, to_play)
        A = model.action_space_size
        discount = self._mcfg["discount_factor"]
        dists, values = roots.get_distributions(), roots.get_values()
        completed = roots.get_children_values(discount, A)
        improved = np.array(roots.get_policies(discount, A))
        output = {}
        for i, env_id in enumerate(ready_env_id):
            mask = np.asarray(action_mask[i])
            _, entropy = select_action(dists[i], temperature=temperature, deterministic=deterministic)
            valid_value = np.where(mask == 1.0, improved[i], 0.0)
            output[env_id] = {
                'action': int(np.argmax([v for v in valid_value])),  # gumbel_muzero.py:591-592
                'visit_count_distributions': dists[i],
                'visit_count_distribution_entropy': entropy,
                'searched_value': values[i],
                'roots_completed_value': np.where(mask == 1.0, np.asarray(completed[i]), 0.0),
                'improved_policy_probs': improved[i],
                'predicted_value': pred_values[i],
                'predicted_policy_logits': policy_logits[i],
            }
        return output

    def _forward_collect(self, data, action_mask=None, temperature=1, to_play=[-1], epsilon=0.25, ready_env_id=None, **kwargs):
        self._collect_mcts_temperature = temperature
        return self._run(self._mcts_collect, self._collect_model, data, action_mask, to_play, True, temperature, False, ready_env_id)

    def forward_collect_rows(self, data, action_mask, rows_out, temperature=1, to_play=[-1], timestep=None, frame_floats=None, epsilon=0.0):
        """The collect forward for a vectorised collector: ``rows_out`` [n, shard.row_width(A, frame_floats, A)] in HBM receives the
        env-step rows (extra block = improved_policy_probs [A]; the action is the arg-max of the improved policy over the legal
        actions, gumbel_muzero.py:591-592), the header block [n, 8 + 3 A] comes back on the host."""
        from .. import shard
        model = self._collect_model
        n, A = data.shape[0], model.action_space_size
        to_play = list(to_play) if len(to_play) == n else [to_play[0]] * n
        mask = np.asarray(action_mask)
        legal_actions = [np.nonzero(mask[j])[0].tolist() for j in range(n)]
        roots = self._roots(n, legal_actions)
        model.initial_inference(data, roots, fetch=False)
        alpha = self._mcfg["root_dirichlet_alpha"]
        noises = [L.rs().dirichlet([alpha] * len(l)).astype(np.float32).tolist() for l in legal_actions]
        roots.prepare_from_inference(self._mcfg["root_noise_weight"], noises, to_play)
        self._mcts_collect.search(roots, model, ("hbm-pool", roots), to_play)
        if frame_floats is None:
            frame_floats = rows_out.shape[1] - shard.HEADER - 3 * A
        hdr, _ = roots.collect_rows(temperature, False, rows_out.data_ptr(), rows_out.shape[1], frame_floats,
                                    discount=self._mcfg["discount_factor"], timestep=timestep,
                                    d_obs_ptr=data.data_ptr() if hasattr(data, "data_ptr") and getattr(data, "is_cuda", False) else None)
        return hdr

    def _forward_eval(self, data, action_mask, to_play=[-1], ready_env_id=None, **kwargs):
        return self._run(self._mcts_eval, self._eval_model, data, action_mask, to_play, False, 1, True, ready_env_id)
