"""Engine-backed counterparts of the vector-observation models (inference graphs): every Linear / BatchNorm1d / LayerNorm /
activation stack runs as the dense MFMA kernel of lightzero_amd/csrc/lz_dense.hip, levelised by lz_mlp.hip.

``MuZeroModelMLP``  lzero/model/muzero_model_mlp.py:13-338 (BASELINE configs[0]: CartPole, latent 128).  Same constructor
keywords as the reference; weights come from a reference-format ``state_dict``.  Pair it with mz_tree.Roots."""
import ctypes

from .. import _lib as L
from .efficientzero_model import EfficientZeroModel


class _EngineModelMLP(EfficientZeroModel):
    _model_type = 2
    _activation = 0       # the model-level default activation: 0 ReLU, 1 GELU(tanh)
    _uses_lstm = False

    def __init__(self, observation_shape=2, action_space_size=6, latent_state_dim=256, lstm_hidden_size=512,
                 reward_support_range=(-300., 301., 1.), value_support_range=(-300., 301., 1.),
                 categorical_distribution=True, state_norm=False, discrete_action_encoding_type='one_hot', norm_type='BN',
                 res_connection_in_dynamics=False, continuous_action_space=False, num_of_sampled_actions=0,
                 sigma_type='conditioned', bound_type=None, engine=None, **kwargs):
        if not categorical_distribution or state_norm:
            raise NotImplementedError("engine model: categorical_distribution=True, state_norm=False")
        if tuple(reward_support_range) != tuple(value_support_range) or value_support_range[2] != 1.:
            raise NotImplementedError("reward and value supports must be equal with step 1")
        if norm_type not in ('BN', 'LN'):
            raise NotImplementedError("norm_type must be 'BN' or 'LN'")
        if continuous_action_space and sigma_type != 'conditioned':
            raise NotImplementedError("sigma_type must be 'conditioned'")
        self.observation_shape = (int(observation_shape),)
        self.action_space_size = int(action_space_size)
        self.latent_state_dim = int(latent_state_dim)
        self.lstm_hidden_size = int(lstm_hidden_size) if self._uses_lstm else 0
        self.continuous_action_space = bool(continuous_action_space)
        self.num_of_sampled_actions = int(num_of_sampled_actions)
        self.value_support_size = int(round((value_support_range[1] - value_support_range[0]) / value_support_range[2]))
        self.reward_support_size = self.value_support_size
        self._policy_width = 2 * self.action_space_size if self.continuous_action_space else self.action_space_size
        self._engine = engine if engine is not None else L.engine_for_new_model()
        enc = 2 if self.continuous_action_space else (0 if discrete_action_encoding_type == 'one_hot' else 1)
        cfg = L.ModelCfg(self._model_type, self.observation_shape[0], 1, 1, self.action_space_size, self.latent_state_dim,
                         self.lstm_hidden_size, 0, 0, self.value_support_size, float(value_support_range[0]), 1e-5, 0,
                         self._activation, 1 if res_connection_in_dynamics else 0, enc, self.num_of_sampled_actions, 0,
                         1 if bound_type == 'tanh' else 0, 1e-5)
        self._create(cfg)

    def initial_inference(self, obs, roots, fetch=True):
        """initial_inference for the batch held by ``roots``; ``obs``: [B, observation_shape] fp32 (device tensor or host
        array).  Same return contract as the convolutional engine models; ``policy_logits`` is [B, A] (or [B, 2 D] =
        (mu | sigma) for continuous actions)."""
        import types
        import numpy as np
        if not self._loaded:
            raise L.LzError("%s: load_state_dict has not been called" % type(self).__name__)
        self._check_owner()
        B = roots.num
        roots._bind_engine(self._engine)
        roots._ensure(self.action_space_size)
        if hasattr(obs, "data_ptr") and getattr(obs, "is_cuda", False):
            if tuple(obs.shape) != (B,) + self.observation_shape or not obs.is_contiguous() or str(obs.dtype) != "torch.float32":
                raise ValueError("obs must be a contiguous float32 [B, observation_shape] tensor")
            import torch
            torch.cuda.current_stream().synchronize()
            L.check(L.lib().lz_initial_inference(roots._h, obs.data_ptr()))
        else:
            arr = np.ascontiguousarray(obs.numpy() if hasattr(obs, "numpy") else obs, dtype=np.float32)
            if arr.shape != (B,) + self.observation_shape:
                raise ValueError("obs must be [B, observation_shape]")
            L.check(L.lib().lz_initial_inference_host(roots._h, arr.reshape(-1)))
        roots._inferred_by = self
        if not fetch:
            return None
        values = np.zeros(B, np.float32)
        logits = np.zeros((B, self._policy_width), np.float32)
        L.check(L.lib().lz_roots_get_root_outputs(roots._h, values, logits.reshape(-1)))
        out = types.SimpleNamespace(value=values, policy_logits=logits, latent_state=("hbm-pool", roots))
        if self._uses_lstm:
            out.value_prefix = [0. for _ in range(B)]
            out.reward_hidden_state = ("hbm-pool", roots)
        else:
            out.reward = [0. for _ in range(B)]
        return out


class MuZeroModelMLP(_EngineModelMLP):
    _model_type = 2
