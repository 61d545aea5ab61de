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
        if value_support_range[2] != 1. or reward_support_range[2] != 1.:
            raise NotImplementedError("supports with step 1")
        self.categorical_distribution = bool(categorical_distribution)
        self.state_norm = bool(state_norm)
        if not self.categorical_distribution:
            # muzero_model_mlp.py:72-77: the value / reward heads have ONE output, the scaled scalar itself (h^-1 applied directly,
            # scaling_transform.py:88-92); the support ranges are not looked at
            reward_support_range = value_support_range = (0., 1., 1.)
        if self.state_norm and int(latent_state_dim) % 4:
            raise NotImplementedError("state_norm=True: latent_state_dim must be a multiple of 4")
        rsize = int(round((reward_support_range[1] - reward_support_range[0]) / reward_support_range[2]))
        vsize = int(round((value_support_range[1] - value_support_range[0]) / value_support_range[2]))
        own_reward_support = tuple(reward_support_range) != tuple(value_support_range)
        if own_reward_support and self._uses_lstm:
            # The reference's EfficientZero drivers transform the value prefix with the VALUE handle (mcts_ctree.py:839-841,
            # mcts_ctree_sampled.py): reward_support_range only sizes the head.  Equal sizes behave exactly like equal supports there
            # (and here); unequal sizes fail in the reference with a shape error.
            if rsize != vsize:
                raise NotImplementedError("EfficientZero: a reward support of another SIZE than the value support fails in the reference's own "
                                          "driver (the value handle is applied to the value prefix, mcts_ctree.py:839-841)")
            own_reward_support = False
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
        self.reward_support_size = rsize
        self._policy_width = 2 * self.action_space_size if self.continuous_action_space else self.action_space_size
        self._engine = engine if engine is not None else L.engine_for_new_model()
        enc = 2 if self.continuous_action_space else (0 if discrete_action_encoding_type == 'one_hot' else 1)
        cfg = L.ModelCfg(self._model_type, self.observation_shape[0], 1, 1, self.action_space_size, self.latent_state_dim,
                         self.lstm_hidden_size, 0, 0, self.value_support_size, float(value_support_range[0]), 1e-5, 0,
                         self._activation, 1 if res_connection_in_dynamics else 0, enc, self.num_of_sampled_actions, 0,
                         1 if bound_type == 'tanh' else 0, 1e-5)
        if own_reward_support:   # MuZeroModelMLP: the MuZero driver transforms rewards with the REWARD handle (mcts_ctree.py:340-346)
            cfg.reward_support_size, cfg.reward_support_min = rsize, float(reward_support_range[0])
        cfg.state_norm = 1 if self.state_norm else 0
        cfg.scalar_heads = 0 if self.categorical_distribution else 1
        self._create(cfg)

    def _latent_shape(self):
        return (self.latent_state_dim,)

    def _new_own_roots(self, B, max_simulations):
        if self._model_type != 4:
            return super()._new_own_roots(B, max_simulations)
        # Sampled EfficientZero: the sampled tree's handle (K actions per node); kept in the model's bounded LRU like every other (_own_roots)
        from ..mcts.ctree.ctree_sampled_efficientzero import ezs_tree
        K = self.num_of_sampled_actions
        return ezs_tree.Roots(B, [[-1] * K] * B, self.action_space_size, K, self.continuous_action_space, max_simulations=max_simulations,
                              engine=self._engine)


class MuZeroModelMLP(_EngineModelMLP):
    _model_type = 2
