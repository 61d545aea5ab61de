"""``SampledEfficientZeroModel``  lzero/model/sampled_efficientzero_model.py:17-485 (the convolutional, pixel-observation Sampled
EfficientZero; inference graph) for DISCRETE action spaces with ``norm_type='BN'`` -- the reference's Atari configuration
(zoo/atari/config/atari_sampled_efficientzero_config.py:36-45: 4 x 64 x 64 observations, ``continuous_action_space=False``,
``num_of_sampled_actions=K``, ``discrete_action_encoding_type='one_hot'``, ``norm_type='BN'``, everything else at the class defaults).

With discrete actions the network has the EfficientZero network's layers -- DownSample tower, dynamics network with one-hot | scalar action
planes and the value-prefix LSTM, prediction network with an MLP policy head over ``action_space_size`` (:610-623) -- under this class's
own defaults: **256** hidden units in the three head MLPs (:29-31) and a MIX of activations that follows from what the constructor passes
on (:177-218): ``activation`` (default **GELU(approximate='tanh')**, :40) reaches the dynamics network only, the prediction network keeps
its own default GELU (:508) and the representation network its default ReLU (common.py:718).  The prediction network's head MLPs are called
``fc_value_head`` / ``fc_policy_head`` (:587-623).  On the engine: the ReLU tower, the chain kernel's GELU instance (an activation code per
layer and per 1x1 convolution), the LSTM's GELU instance and the heads as dense layers + row finishers (lz_search.hip::wide_heads).  What differs from EfficientZero
in the search is the tree: every node holds K actions sampled from its policy (mcts_ctree_sampled.py, ctree_sampled_efficientzero), so
this model pairs with ``ezs_tree.Roots(root_num, legal, action_space_size, K, False)`` (lz_model_cfg.num_of_sampled_actions > 0 on
model_type 0).

Refused with the reason: ``continuous_action_space=True`` (ReparameterizationHead on conv features, action-value planes in the dynamics
input: zoo/dmc2gym/config/dmc2gym_pixels_sez_config.py) and ``norm_type='LN'`` (the class default; the Atari configuration sets 'BN')."""
from .efficientzero_model import EfficientZeroModel


def _act_code(activation):
    """None -> the class default GELU(tanh); a torch module (nn.GELU(approximate='tanh') | nn.ReLU) or its name"""
    if activation is None:
        return 1
    name = activation if isinstance(activation, str) else type(activation).__name__
    name = {"gelu": "GELU", "relu": "ReLU"}.get(name.lower(), name)
    if name == "GELU":
        if not isinstance(activation, str) and getattr(activation, "approximate", "tanh") != "tanh":
            raise NotImplementedError("GELU(approximate='tanh') only (the reference's default)")
        return 1
    if name == "ReLU":
        return 0
    raise NotImplementedError("activation must be GELU(approximate='tanh') or ReLU")


class SampledEfficientZeroModel(EfficientZeroModel):
    _model_type = 0
    _uses_lstm = True

    def __init__(self, observation_shape=(4, 64, 64), action_space_size=6, num_of_sampled_actions=6, continuous_action_space=False,
                 norm_type='LN', activation=None, reward_head_hidden_channels=(256,), value_head_hidden_channels=(256,),
                 policy_head_hidden_channels=(256,), downsample=False, sigma_type='conditioned', fixed_sigma_value=0.3, bound_type=None,
                 **kwargs):
        if continuous_action_space:
            raise NotImplementedError("engine SampledEfficientZeroModel (conv): discrete action spaces only (continuous actions: SampledEfficientZeroModelMLP)")
        if norm_type != 'BN':
            raise NotImplementedError("engine SampledEfficientZeroModel (conv): norm_type='BN' (the reference's Atari configuration); the class default 'LN' has no kernels")
        if not downsample or tuple(observation_shape)[0] != 4:
            raise NotImplementedError("engine SampledEfficientZeroModel (conv): downsample=True on 4-channel 64x64 | 96x96 observations")
        if not 1 <= int(num_of_sampled_actions) <= 64:
            raise NotImplementedError("num_of_sampled_actions must be in [1, 64]")
        self.continuous_action_space = False
        self.num_of_sampled_actions = int(num_of_sampled_actions)
        self._activation_code = _act_code(activation)
        self._policy_width = int(action_space_size)
        super().__init__(observation_shape=observation_shape, action_space_size=action_space_size, norm_type='BN', downsample=True,
                         reward_head_hidden_channels=reward_head_hidden_channels, value_head_hidden_channels=value_head_hidden_channels,
                         policy_head_hidden_channels=policy_head_hidden_channels, **kwargs)

    def _create(self, cfg):
        cfg.num_of_sampled_actions = self.num_of_sampled_actions   # model_type 0 + K > 0: searched by the sampled tree
        cfg.activation = self._activation_code
        super()._create(cfg)

    def load_state_dict(self, state_dict, strict=True):
        """the reference's key names (prediction_network.fc_value_head.* / fc_policy_head.*) -> the EfficientZero network's"""
        from .efficientzero_model import unwrap_checkpoint
        state_dict = unwrap_checkpoint(state_dict)   # a whole checkpoint {'model', 'target_model', 'optimizer'}: rename INSIDE its 'model' dict
        ren = {}
        for k, v in state_dict.items():
            k2 = k.replace("prediction_network.fc_value_head.", "prediction_network.fc_value.") \
                  .replace("prediction_network.fc_policy_head.", "prediction_network.fc_policy.")
            ren[k2] = v
        return super().load_state_dict(ren, strict)

    def _new_own_roots(self, B, max_simulations):
        # the sampled tree's handle (K actions per node, discrete action space); kept in the model's bounded LRU (_own_roots)
        from ..mcts.ctree.ctree_sampled_efficientzero import ezs_tree
        K, A = self.num_of_sampled_actions, self.action_space_size
        return ezs_tree.Roots(B, [list(range(A))] * B, A, K, False, max_simulations=max_simulations, engine=self._engine)
