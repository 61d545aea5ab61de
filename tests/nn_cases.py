"""Shared definition of the network parity cases: the constructor keywords (identical for the reference class, the oracle
restatement and the engine model), the weight seed and the seeded inputs.  Used by
  * tests/test_torch_models_vs_reference.py  (CPU, here): oracle restatement == the reference's own modules, bit for bit
  * tests/golden/make_golden_nn.py           (CPU, here): reference-module outputs -> tests/golden/nn_*.npz
  * tests/test_nn_golden_gpu.py              (GPU box):   HIP kernels vs those files
Inputs come from numpy's PCG64 (identical on every machine), never from a torch generator."""
import numpy as np

CASES = {
    # BASELINE configs[1]: EfficientZero Atari, 96x96x4 -> 6x6 latent
    "ez_atari96": dict(family="ez", kw=dict(observation_shape=(4, 96, 96), action_space_size=6, downsample=True), B=6, seed=11),
    # the shipped Atari config: 64x64 -> 8x8 latent, supports (-50, 51, 1)
    "ez_atari64": dict(family="ez", kw=dict(observation_shape=(4, 64, 64), action_space_size=6, downsample=True,
                                            reward_support_range=(-50., 51., 1.), value_support_range=(-50., 51., 1.)), B=5, seed=12),
    # BASELINE configs[2]: MuZero Atari
    "mz_atari96": dict(family="mz", kw=dict(observation_shape=(4, 96, 96), action_space_size=4, downsample=True), B=6, seed=13),
    # BASELINE configs[3]: Go 9x9 board, 82 actions, no downsample
    "mz_go9": dict(family="mz", kw=dict(observation_shape=(17, 9, 9), action_space_size=82, downsample=False), B=5, seed=14),
    # the reference's narrow board-game models: gomoku (zoo/board_games/gomoku/config/gomoku_muzero_bot_mode_config.py:36-45: 32 channels,
    # 6x6 board, supports (-10, 11, 1)) and tictactoe (tictactoe_muzero_bot_mode_config.py:26-39: 16 channels, 3x3, head hidden [8])
    "mz_gomoku_c32": dict(family="mz", kw=dict(observation_shape=(3, 6, 6), action_space_size=36, downsample=False, num_channels=32,
                                               reward_support_range=(-10., 11., 1.), value_support_range=(-10., 11., 1.)), B=7, seed=19),
    "mz_tictactoe_c16": dict(family="mz", kw=dict(observation_shape=(3, 3, 3), action_space_size=9, downsample=False, num_channels=16,
                                                  reward_head_hidden_channels=[8], value_head_hidden_channels=[8], policy_head_hidden_channels=[8],
                                                  reward_support_range=(-10., 11., 1.), value_support_range=(-10., 11., 1.)), B=7, seed=20),
    # num_res_blocks > 1 (RepresentationNetwork / DynamicsNetwork / PredictionNetwork are generic in it, common.py:706-787)
    "ez_atari96_rb2": dict(family="ez", kw=dict(observation_shape=(4, 96, 96), action_space_size=6, downsample=True, num_res_blocks=2), B=4, seed=21),
    "mz_connect4_rb3": dict(family="mz", kw=dict(observation_shape=(3, 6, 7), action_space_size=7, downsample=False, num_res_blocks=3), B=5, seed=22),
    # discrete_action_encoding_type='not_one_hot' (one plane holding action / action_space_size; the reference ships it for chinese chess,
    # zoo/board_games/chinese_chess/config/chinese_chess_muzero_bot_mode_config.py) and unequal reward / value supports (MuZero only: the
    # reference's EfficientZero driver sends the value prefix through the VALUE handle, mcts_ctree.py:839-841)
    "ez_atari96_noh": dict(family="ez", kw=dict(observation_shape=(4, 96, 96), action_space_size=6, downsample=True,
                                                discrete_action_encoding_type='not_one_hot'), B=5, seed=23),
    "mz_board_noh_supports": dict(family="mz", kw=dict(observation_shape=(5, 6, 6), action_space_size=36, downsample=False,
                                                       discrete_action_encoding_type='not_one_hot',
                                                       reward_support_range=(-2., 3., 1.), value_support_range=(-10., 11., 1.)), B=6, seed=24),
    # BASELINE configs[0]: CartPole MuZeroModelMLP
    "mz_mlp_cartpole": dict(family="mz_mlp", kw=dict(observation_shape=4, action_space_size=2, latent_state_dim=128), B=8, seed=15),
    # MuZeroModelMLP with a reward support of its own (muzero_model_mlp.py:73-74; the MuZero driver transforms rewards with the REWARD
    # handle, mcts_ctree.py:60-63,340-346)
    "mz_mlp_supports": dict(family="mz_mlp", kw=dict(observation_shape=6, action_space_size=3, latent_state_dim=128,
                                                     reward_support_range=(-5., 6., 1.), value_support_range=(-20., 21., 1.)), B=7, seed=34),
    "ez_mlp": dict(family="ez_mlp", kw=dict(observation_shape=6, action_space_size=3, lstm_hidden_size=128, latent_state_dim=128), B=8, seed=16),
    # state_norm=True (the latent renormalised to [0, 1] after the representation and after the dynamics network, lzero/model/utils.py:242-271)
    # and categorical_distribution=False (one-output value / reward heads, h^-1 on the scalar: scaling_transform.py:88-92) -- the two
    # constructor keywords of the MLP models the engine refused until round 6 (muzero_model_mlp.py:30,33)
    "mz_mlp_statenorm": dict(family="mz_mlp", kw=dict(observation_shape=6, action_space_size=3, latent_state_dim=128, state_norm=True), B=7, seed=35),
    "ez_mlp_statenorm_res": dict(family="ez_mlp", kw=dict(observation_shape=8, action_space_size=4, lstm_hidden_size=256, latent_state_dim=256, state_norm=True,
                                                          res_connection_in_dynamics=True, norm_type='LN'), B=9, seed=36),
    "mz_mlp_scalar": dict(family="mz_mlp", kw=dict(observation_shape=5, action_space_size=3, latent_state_dim=128, categorical_distribution=False), B=6, seed=37),
    "ez_mlp_scalar_statenorm": dict(family="ez_mlp", kw=dict(observation_shape=6, action_space_size=3, lstm_hidden_size=128, latent_state_dim=128,
                                                             categorical_distribution=False, state_norm=True), B=8, seed=38),
    "sez_mlp_statenorm": dict(family="sez_mlp", kw=dict(observation_shape=5, action_space_size=2, num_of_sampled_actions=6, continuous_action_space=True,
                                                        lstm_hidden_size=256, latent_state_dim=256, state_norm=True), B=7, seed=39),
    # 8x8 boards without downsample (64 channels): the 8x8 Winograd chain of the 64x64 Atari latents serves them
    "mz_board8": dict(family="mz", kw=dict(observation_shape=(3, 8, 8), action_space_size=65, downsample=False, num_res_blocks=2), B=6, seed=28),
    "ez_board8": dict(family="ez", kw=dict(observation_shape=(5, 8, 8), action_space_size=64, downsample=False), B=5, seed=29),
    # TicTacToe EfficientZero (zoo/board_games/tictactoe/config/tictactoe_efficientzero_bot_mode_config.py): the 16-channel model with the LSTM
    "ez_tictactoe_c16": dict(family="ez", kw=dict(observation_shape=(3, 3, 3), action_space_size=9, downsample=False, num_channels=16,
                                                  reward_head_hidden_channels=[8], value_head_hidden_channels=[8], policy_head_hidden_channels=[8],
                                                  reward_support_range=(-10., 11., 1.), value_support_range=(-10., 11., 1.)), B=7, seed=31),
    # 2048 (zoo/game_2048/config/muzero_2048_config.py: observation (16, 4, 4), 4 actions)
    "mz_2048": dict(family="mz", kw=dict(observation_shape=(16, 4, 4), action_space_size=4, downsample=False), B=7, seed=30),
    # the sizes of the reference's LunarLander / BipedalWalker / MuJoCo / MiniGrid configs: latent 256 + LSTM 256
    # (zoo/box2d/lunarlander/config/lunarlander_disc_efficientzero_config.py, zoo/mujoco/config/mujoco_sampled_efficientzero_config.py)
    "ez_mlp_lunarlander": dict(family="ez_mlp", kw=dict(observation_shape=8, action_space_size=4, lstm_hidden_size=256, latent_state_dim=256), B=9, seed=25),
    "sez_mlp_mujoco": dict(family="sez_mlp", kw=dict(observation_shape=11, action_space_size=3, num_of_sampled_actions=20, continuous_action_space=True,
                                                     lstm_hidden_size=256, latent_state_dim=256), B=7, seed=26),
    "ez_mlp_128_256": dict(family="ez_mlp", kw=dict(observation_shape=5, action_space_size=3, lstm_hidden_size=256, latent_state_dim=128), B=6, seed=27),
    # MiniGrid (zoo/minigrid/config/minigrid_muzero_config.py / minigrid_efficientzero_config.py): 2835 observation features
    "mz_mlp_minigrid": dict(family="mz_mlp", kw=dict(observation_shape=2835, action_space_size=7, latent_state_dim=512), B=6, seed=32),
    "ez_mlp_minigrid": dict(family="ez_mlp", kw=dict(observation_shape=2835, action_space_size=7, lstm_hidden_size=256, latent_state_dim=256), B=5, seed=33),
    # BASELINE configs[4]: Sampled EfficientZero, continuous actions, K = 20
    "sez_mlp_cont": dict(family="sez_mlp", kw=dict(observation_shape=5, action_space_size=1, num_of_sampled_actions=20,
                                                   continuous_action_space=True), B=8, seed=17),
    "sez_mlp_disc": dict(family="sez_mlp", kw=dict(observation_shape=6, action_space_size=5, num_of_sampled_actions=3,
                                                   continuous_action_space=False), B=8, seed=18),
    # the convolutional Sampled EfficientZero with discrete actions, the reference's Atari configuration as shipped
    # (zoo/atari/config/atari_sampled_efficientzero_config.py:36-45: 4 x 64 x 64 observations, K sampled actions, one_hot, norm_type='BN'; the class
    # defaults stay: GELU(tanh) activations, 256-wide head MLPs): the EfficientZero layers, searched by the sampled tree
    "sez_atari64": dict(family="sez", kw=dict(observation_shape=(4, 64, 64), action_space_size=6, num_of_sampled_actions=5, downsample=True,
                                              continuous_action_space=False, norm_type='BN'), B=5, seed=35),
    "sez_atari96_relu32": dict(family="sez", kw=dict(observation_shape=(4, 96, 96), action_space_size=9, num_of_sampled_actions=4, downsample=True,
                                                     continuous_action_space=False, norm_type='BN', activation="relu",
                                                     reward_head_hidden_channels=[32], value_head_hidden_channels=[32],
                                                     policy_head_hidden_channels=[32]), B=4, seed=36),
}
# Configurations the ENGINE models do not take (128 channels on a 10 x 9 board) but whose torch restatement drives a committed golden: the
# Chinese chess preset behind tests/golden/driver_mz_xiangqi_2p_b4.npz (foreign torch model + the device tree for 2086 actions).  Only the
# pin of oracle/torch_models.py to the reference module runs on them (tests/test_torch_models_vs_reference.py).
ORACLE_ONLY_CASES = {
    "mz_xiangqi": dict(family="mz", kw=dict(observation_shape=(57, 10, 9), action_space_size=2086, num_res_blocks=6, num_channels=128,
                                             reward_head_hidden_channels=[128], value_head_hidden_channels=[128],
                                             policy_head_hidden_channels=[256], downsample=False,
                                             reward_support_range=(-1., 1., 1.), value_support_range=(-1., 1., 1.),
                                             discrete_action_encoding_type='not_one_hot'), B=3, seed=63),
}
STEPS = 3  # recurrent inferences chained after the initial one (teacher-forced on the reference's own states)


def oracle_class(tm, family):
    return {"ez": tm.EfficientZeroModel, "mz": tm.MuZeroModel, "mz_mlp": tm.MuZeroModelMLP, "ez_mlp": tm.EfficientZeroModelMLP,
            "sez_mlp": tm.SampledEfficientZeroModelMLP, "sez": tm.SampledEfficientZeroModel}[family]


def reference_class(ref, family):
    return {"ez": ref.efficientzero_model.EfficientZeroModel, "mz": ref.muzero_model.MuZeroModel,
            "mz_mlp": ref.muzero_model_mlp.MuZeroModelMLP, "ez_mlp": ref.efficientzero_model_mlp.EfficientZeroModelMLP,
            "sez_mlp": ref.sampled_efficientzero_model_mlp.SampledEfficientZeroModelMLP,
            "sez": ref.sampled_efficientzero_model.SampledEfficientZeroModel}[family]


def engine_class(family):
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    from lightzero_amd.model.muzero_model import MuZeroModel
    from lightzero_amd.model.muzero_model_mlp import MuZeroModelMLP
    from lightzero_amd.model.efficientzero_model_mlp import EfficientZeroModelMLP
    from lightzero_amd.model.sampled_efficientzero_model_mlp import SampledEfficientZeroModelMLP
    from lightzero_amd.model.sampled_efficientzero_model import SampledEfficientZeroModel
    return {"ez": EfficientZeroModel, "mz": MuZeroModel, "mz_mlp": MuZeroModelMLP, "ez_mlp": EfficientZeroModelMLP,
            "sez_mlp": SampledEfficientZeroModelMLP, "sez": SampledEfficientZeroModel}[family]


def reference_kwargs(case):
    """the reference constructors take one more switch the restatement has no use for"""
    kw = dict(case["kw"])
    if case["family"] in ("ez", "mz"):
        kw["self_supervised_learning_loss"] = False
    else:
        kw["self_supervised_learning_loss"] = False
    if kw.get("activation") == "relu":   # (the cases name it; the reference takes the module)
        import torch.nn as nn
        kw["activation"] = nn.ReLU(inplace=True)
    return kw


def has_lstm(family):
    return family in ("ez", "ez_mlp", "sez_mlp", "sez")


def inputs(case):
    """seeded observation batch and the action sequence of the chained recurrent steps"""
    rng = np.random.default_rng(case["seed"])
    kw, B = case["kw"], case["B"]
    shape = kw["observation_shape"]
    if isinstance(shape, int):
        obs = rng.standard_normal((B, shape)).astype(np.float32)
    else:
        obs = rng.random((B,) + tuple(shape), dtype=np.float32)
    if case["family"] == "sez_mlp" and kw.get("continuous_action_space", True):
        actions = np.tanh(rng.standard_normal((STEPS, B, kw["action_space_size"]))).astype(np.float32)
    else:
        actions = rng.integers(0, kw["action_space_size"], size=(STEPS, B)).astype(np.int64)
    return obs, actions
