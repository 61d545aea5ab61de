"""Collect / eval halves of lzero/policy/sampled_efficientzero.py::SampledEfficientZeroPolicy (``_forward_collect``
:797-935, ``_forward_eval`` :937-1060; continuous action spaces) on the HBM-resident Sampled-EfficientZero tree.

``model`` is either the engine model (lightzero_amd.model.sampled_efficientzero_model_mlp.SampledEfficientZeroModelMLP:
the whole search, network and action sampling included, stays on the device) or any torch module with the reference
model's inference contract (the reference loop with the device tree).  Same arguments and per-env output dict as the
reference (incl. ``root_sampled_actions``)."""
import numpy as np

from .. import _lib as L
from ..mcts.tree_search.mcts_ctree import SampledEfficientZeroMCTSCtree as MCTSCtree, _inverse_scalar_transform
from .efficientzero import _g, _mcts_seed
from .utils import CheckpointIngest, select_action


class SampledEfficientZeroPolicy(CheckpointIngest):
    def __init__(self, cfg, model):
        self._cfg = cfg
        self._collect_model = model
        self._eval_model = model
        mc = _g(cfg, "model", {}) or {}
        self._A = int(_g(mc, "action_space_size"))
        self._K = int(_g(mc, "num_of_sampled_actions", 20))
        self._continuous = bool(_g(mc, "continuous_action_space", True))
        self._support_min = float(_g(mc, "value_support_range", (-300., 301., 1.))[0])
        self._mcfg = dict(num_simulations=_g(cfg, "num_simulations", 50), discount_factor=_g(cfg, "discount_factor", 0.997),
                          lstm_horizon_len=_g(cfg, "lstm_horizon_len", 5), pb_c_base=_g(cfg, "pb_c_base", 19652),
                          pb_c_init=_g(cfg, "pb_c_init", 1.25), value_delta_max=_g(cfg, "value_delta_max", 0.01),
                          root_dirichlet_alpha=_g(cfg, "root_dirichlet_alpha", 0.3),
                          root_noise_weight=_g(cfg, "root_noise_weight", 0.25),
                          env_type=_g(cfg, "env_type", "not_board_games"), device=_g(cfg, "device", "cpu"), model=mc)
        self._mcts_collect = MCTSCtree(self._mcfg)
        self._mcts_eval = MCTSCtree(self._mcfg)
        self._collect_mcts_temperature = 1.
        self._roots_cache = {}
        self._tiebreak = {"random": 1, "first": 0}[_g(cfg, "mcts_tiebreak", "random")]

    def forward(self, *args, **kwargs):
        return self._forward_collect(*args, **kwargs)

    def _roots(self, n):
        # the reference builds a fresh Roots per forward (sampled_efficientzero.py:876); prepare() re-arms the same pools
        roots = self._roots_cache.get(n)
        if roots is None:
            # the reference passes the action mask's indices for discrete spaces; its CRoots ignores them (cnode.cpp:653-659)
            legal_actions = [[-1 for _ in range(self._K)] for _ in range(n)]
            roots = MCTSCtree.roots(n, legal_actions, self._A, self._K, self._continuous, max_simulations=int(self._mcfg["num_simulations"]))
            roots.set_tiebreak(self._tiebreak, seed=_mcts_seed(self._cfg))
            self._roots_cache[n] = roots
        return roots

    def _run(self, mcts, model, data, to_play, noise):
        n = data.shape[0]
        roots = self._roots(n)
        alpha = self._mcfg["root_dirichlet_alpha"]
        noises = [L.rs().dirichlet([alpha] * self._K).astype(np.float32).tolist() for _ in range(n)] if noise else None
        if getattr(model, "_is_lz_engine_model", False):
            out = model.initial_inference(data, roots)
            pred_values, policy_logits = out.value, out.policy_logits.tolist()
            if noise:
                roots.prepare_from_inference(self._mcfg["root_noise_weight"], noises, to_play)
            else:
                roots.prepare_from_inference_no_noise(to_play)
            mcts.search(roots, model, out.latent_state, out.reward_hidden_state, to_play)
        else:
            import torch
            with torch.no_grad():
                model.eval()
                out = model.initial_inference(data)
                pred_values = _inverse_scalar_transform(out.value, self._support_min)
                latent_state_roots = out.latent_state.detach().cpu().numpy()
                reward_hidden_state_roots = (out.reward_hidden_state[0].detach().cpu().numpy(),
                                             out.reward_hidden_state[1].detach().cpu().numpy())
                policy_logits = out.policy_logits.detach().cpu().numpy().tolist()
            if noise:
                roots.prepare(self._mcfg["root_noise_weight"], noises, list(out.value_prefix), policy_logits, to_play)
            else:
                roots.prepare_no_noise(list(out.value_prefix), policy_logits, to_play)
            mcts.search(roots, model, latent_state_roots, reward_hidden_state_roots, to_play)
        return roots, pred_values, policy_logits

    def _output(self, roots, pred_values, policy_logits, ready_env_id, temperature, deterministic):
        roots_visit_count_distributions = roots.get_distributions()
        roots_values = roots.get_values()
        roots_sampled_actions = roots.get_sampled_actions()
        output = {}
        for i, env_id in enumerate(ready_env_id):
            distributions, value = roots_visit_count_distributions[i], roots_values[i]
            root_sampled_actions = np.array([a for a in roots_sampled_actions[i]])
            idx, entropy = select_action(distributions, temperature=temperature, deterministic=deterministic)
            action = np.array(roots_sampled_actions[i][idx])
            if not self._continuous:  # sampled_efficientzero.py:908-913
                action = int(action) if len(action.shape) == 0 else int(action[0])
            output[env_id] = {
                'action': action,
                'visit_count_distributions': distributions,
                'root_sampled_actions': root_sampled_actions,
                'visit_count_distribution_entropy': entropy,
                'searched_value': value,
                'predicted_value': pred_values[i],
                'predicted_policy_logits': policy_logits[i],
            }
        return output

    def _forward_collect(self, data, action_mask=None, temperature=1, to_play=[-1], epsilon=0.25, ready_env_id=None,
                         **kwargs):
        self._collect_mcts_temperature = temperature
        n = data.shape[0]
        if ready_env_id is None:
            ready_env_id = np.arange(n)
        to_play = list(to_play) if len(to_play) == n else [to_play[0]] * n
        roots, pred_values, policy_logits = self._run(self._mcts_collect, self._collect_model, data, to_play, True)
        return self._output(roots, pred_values, policy_logits, ready_env_id, self._collect_mcts_temperature, False)

    def forward_collect_rows(self, data, action_mask, rows_out, temperature=1, to_play=[-1], timestep=None, frame_floats=None, epsilon=0.0):
        """The collect forward for a vectorised collector (EfficientZeroPolicy.forward_collect_rows for this family): ``rows_out``
        [n, shard.row_width(K, frame_floats, K * D)] in HBM receives the env-step rows -- visit block and mask over the K sampled
        actions, the extra block = root_sampled_actions [K][D], word 0 = the POSITION of the chosen action among them -- and the
        header block [n, 8 + 2 K + K D] comes back on the host (engine model only)."""
        from .. import shard
        n = data.shape[0]
        to_play = list(to_play) if len(to_play) == n else [to_play[0]] * n
        self._collect_mcts_temperature = temperature
        if not getattr(self._collect_model, "_is_lz_engine_model", False):
            raise NotImplementedError("forward_collect_rows needs the engine model; any other model is served by _forward_collect")
        roots, _, _ = self._run(self._mcts_collect, self._collect_model, data, to_play, True)
        E = self._K * (self._A if self._continuous else 1)
        if frame_floats is None:
            frame_floats = rows_out.shape[1] - shard.HEADER - 2 * self._K - E
        hdr, _ = roots.collect_rows(temperature, False, rows_out.data_ptr(), rows_out.shape[1], frame_floats, timestep=timestep)
        return hdr

    def _forward_eval(self, data, action_mask=None, to_play=[-1], ready_env_id=None, **kwargs):
        n = data.shape[0]
        if ready_env_id is None:
            ready_env_id = np.arange(n)
        to_play = list(to_play) if len(to_play) == n else [to_play[0]] * n
        roots, pred_values, policy_logits = self._run(self._mcts_eval, self._eval_model, data, to_play, False)
        return self._output(roots, pred_values, policy_logits, ready_env_id, 1, True)
