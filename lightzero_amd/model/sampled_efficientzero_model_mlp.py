"""``SampledEfficientZeroModelMLP``  lzero/model/sampled_efficientzero_model_mlp.py (inference graph, continuous action
spaces, BASELINE configs[4]; or discrete ones with an MLP policy head): LayerNorm + GELU(tanh) stacks, value-prefix LSTM,
ReparameterizationHead (mu | sigma).
Pair it with ezs_tree.Roots.  See muzero_model_mlp.py for the engine mechanics."""
from .muzero_model_mlp import _EngineModelMLP


class SampledEfficientZeroModelMLP(_EngineModelMLP):
    _model_type = 4
    _activation = 1
    _uses_lstm = True

    def __init__(self, observation_shape=2, action_space_size=6, latent_state_dim=256, lstm_hidden_size=512,
                 continuous_action_space=False, num_of_sampled_actions=6, norm_type='LN', res_connection_in_dynamics=True,
                 **kwargs):
        if not continuous_action_space and int(num_of_sampled_actions) > int(action_space_size):
            # zoo/memory/config/memory_sampled_efficientzero_config.py ships exactly this (4 actions, K = 5)
            raise NotImplementedError("discrete Sampled EfficientZero with num_of_sampled_actions (%d) > action_space_size (%d): the reference's "
                                      "expand takes the first K entries of a sorted list of action_space_size entries "
                                      "(ctree_sampled_efficientzero/lib/cnode.cpp:403-407) -- it reads past the list, undefined behaviour there, "
                                      "refused here" % (int(num_of_sampled_actions), int(action_space_size)))
        super().__init__(observation_shape=observation_shape, action_space_size=action_space_size,
                         latent_state_dim=latent_state_dim, lstm_hidden_size=lstm_hidden_size,
                         continuous_action_space=bool(continuous_action_space), num_of_sampled_actions=num_of_sampled_actions,
                         norm_type=norm_type, res_connection_in_dynamics=res_connection_in_dynamics, **kwargs)
