"""TEST INFRASTRUCTURE ONLY -- torch fp32 restatement of the reference networks on the hot path.

The reference modules cannot be imported here (they import ``ding`` / ``easydict``, not installed),
so the inference graphs are restated with the reference's parameter names (a LightZero checkpoint's
``state_dict`` keys line up one-to-one):

* ``EfficientZeroModel``     lzero/model/efficientzero_model.py:20-382 (initial/recurrent inference),
                             ``DynamicsNetwork`` :427-569
* ``RepresentationNetwork``  lzero/model/common.py:706-787, ``DownSample`` :266-365
* ``PredictionNetwork``      lzero/model/common.py:1081-1216, ``MLP_V2`` :28-98
* DI-engine pieces (un-vendored third party, ``DI-engine>=0.5.3``, requirements.txt:1), restated from
  its published source: ``ding.torch_utils.ResBlock`` (res_type 'basic': conv3x3-norm-act, conv3x3-norm,
  +identity, act; 'downsample': conv3x3/s2-norm-act, conv3x3-norm, identity = conv3x3/s2 without norm)
  built from ``conv2d_block`` = nn.Sequential(conv, norm, act) and ``ding.torch_utils.MLP`` =
  nn.Sequential(Linear, norm, act, ..., Linear).
* ``InverseScalarTransform`` lzero/policy/scaling_transform.py:64-92

The reference's own tests assert shapes only (lzero/model/tests/test_efficientzero_model.py:89-127).  PIN: every class here
is checked BIT-EQUAL against the reference's own module of the same name, imported from /root/reference as it lies with a stub
``ding`` (tests/test_torch_models_vs_reference.py; the residual restatement is DI-engine's ResBlock / MLP /
ReparameterizationHead in tests/ref_stubs), ``InverseScalarTransform`` against lzero/policy/scaling_transform.py directly
(incl. the reference's own check lzero/policy/tests/test_scaling_transform.py:7-19), and against the committed outputs of those
reference modules (tests/golden/nn_*.npz, tests/test_nn_golden_cpu.py) on machines without /root/reference.
"""
import math
from collections import namedtuple

import torch
import torch.nn as nn

EZNetworkOutput = namedtuple("EZNetworkOutput", "value value_prefix policy_logits latent_state reward_hidden_state")
MZNetworkOutput = namedtuple("MZNetworkOutput", "value reward policy_logits latent_state")


def conv2d_block(cin, cout, k, stride, pad, activation, norm, bias):
    layers = [nn.Conv2d(cin, cout, k, stride, pad, bias=bias)]
    if norm:
        layers.append(nn.BatchNorm2d(cout))
    if activation is not None:
        layers.append(activation)
    return nn.Sequential(*layers)


class ResBlock(nn.Module):
    """ding.torch_utils.ResBlock, res_type in {'basic', 'downsample'}, norm_type='BN'."""

    def __init__(self, in_channels, out_channels=None, res_type="basic", bias=False):
        super().__init__()
        out_channels = out_channels or in_channels
        self.act = nn.ReLU(inplace=True)
        self.res_type = res_type
        stride = 2 if res_type == "downsample" else 1
        self.conv1 = conv2d_block(in_channels, out_channels, 3, stride, 1, self.act, True, bias)
        self.conv2 = conv2d_block(out_channels, out_channels, 3, 1, 1, None, True, bias)
        if res_type == "downsample":
            self.conv3 = conv2d_block(in_channels, out_channels, 3, 2, 1, None, False, bias)

    def forward(self, x):
        identity = x
        x = self.conv1(x)
        x = self.conv2(x)
        if self.res_type == "downsample":
            identity = self.conv3(identity)
        return self.act(x + identity)


def mlp(cin, hidden, cout):
    """ding MLP(layer_num=2, norm_type='BN', output_activation=False, output_norm=False) == MLP_V2 with one
    hidden layer: Linear - BN1d - ReLU - Linear (indices 0,1,2,3)."""
    return nn.Sequential(nn.Linear(cin, hidden), nn.BatchNorm1d(hidden), nn.ReLU(inplace=True), nn.Linear(hidden, cout))


class DownSample(nn.Module):  # common.py:266-365
    def __init__(self, observation_shape, out_channels):
        super().__init__()
        self.observation_shape = observation_shape
        self.activation = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(observation_shape[0], out_channels // 2, 3, stride=2, padding=1, bias=False)
        self.norm1 = nn.BatchNorm2d(out_channels // 2)
        self.resblocks1 = nn.ModuleList([ResBlock(out_channels // 2)])
        self.downsample_block = ResBlock(out_channels // 2, out_channels, res_type="downsample")
        self.resblocks2 = nn.ModuleList([ResBlock(out_channels)])
        self.pooling1 = nn.AvgPool2d(kernel_size=3, stride=2, padding=1)
        self.resblocks3 = nn.ModuleList([ResBlock(out_channels)])
        self.pooling2 = nn.AvgPool2d(kernel_size=3, stride=2, padding=1)

    def forward(self, x):
        x = self.activation(self.norm1(self.conv1(x)))
        for b in self.resblocks1:
            x = b(x)
        x = self.downsample_block(x)
        for b in self.resblocks2:
            x = b(x)
        x = self.pooling1(x)
        for b in self.resblocks3:
            x = b(x)
        if self.observation_shape[1] == 64:
            return x
        return self.pooling2(x)


class RepresentationNetwork(nn.Module):  # common.py:706-787
    def __init__(self, observation_shape, num_res_blocks, num_channels, downsample):
        super().__init__()
        self.downsample = downsample
        self.activation = nn.ReLU(inplace=True)
        if downsample:
            self.downsample_net = DownSample(observation_shape, num_channels)
        else:
            self.conv = nn.Conv2d(observation_shape[0], num_channels, 3, 1, 1, bias=False)
            self.norm = nn.BatchNorm2d(num_channels)
        self.resblocks = nn.ModuleList([ResBlock(num_channels) for _ in range(num_res_blocks)])

    def forward(self, x):
        if self.downsample:
            x = self.downsample_net(x)
        else:
            x = self.activation(self.norm(self.conv(x)))
        for b in self.resblocks:
            x = b(x)
        return x


class PredictionNetwork(nn.Module):  # common.py:1081-1216
    def __init__(self, action_space_size, num_res_blocks, num_channels, value_head_channels, policy_head_channels,
                 value_hidden, policy_hidden, support_size, flat_value, flat_policy):
        super().__init__()
        self.resblocks = nn.ModuleList([ResBlock(num_channels) for _ in range(num_res_blocks)])
        self.conv1x1_value = nn.Conv2d(num_channels, value_head_channels, 1)
        self.conv1x1_policy = nn.Conv2d(num_channels, policy_head_channels, 1)
        self.norm_value = nn.BatchNorm2d(value_head_channels)
        self.norm_policy = nn.BatchNorm2d(policy_head_channels)
        self.flat_value, self.flat_policy = flat_value, flat_policy
        self.activation = nn.ReLU(inplace=True)
        self.fc_value = mlp(flat_value, value_hidden, support_size)
        self.fc_policy = mlp(flat_policy, policy_hidden, action_space_size)

    def forward(self, latent_state):
        for b in self.resblocks:
            latent_state = b(latent_state)
        value = self.activation(self.norm_value(self.conv1x1_value(latent_state)))
        policy = self.activation(self.norm_policy(self.conv1x1_policy(latent_state)))
        value = self.fc_value(value.reshape(-1, self.flat_value))
        policy = self.fc_policy(policy.reshape(-1, self.flat_policy))
        return policy, value


def _action_encoding(model, latent_state, action):
    """efficientzero_model.py:335-369 / muzero_model.py (same code): one-hot planes, or ONE plane holding action / action_space_size"""
    if model.discrete_action_encoding_type == 'one_hot':
        if len(action.shape) == 1:
            action = action.unsqueeze(-1)
        action_one_hot = torch.zeros(action.shape[0], model.action_space_size, device=action.device)
        action_one_hot.scatter_(1, action.long(), 1)
        return action_one_hot.unsqueeze(-1).unsqueeze(-1).expand(
            latent_state.shape[0], model.action_space_size, latent_state.shape[2], latent_state.shape[3])
    if len(action.shape) == 2:
        action = action.unsqueeze(-1).unsqueeze(-1)
    elif len(action.shape) == 1:
        action = action.unsqueeze(-1).unsqueeze(-1).unsqueeze(-1)
    return action.expand(latent_state.shape[0], 1, latent_state.shape[2], latent_state.shape[3]) / model.action_space_size


class EZDynamicsNetwork(nn.Module):  # efficientzero_model.py:427-569
    def __init__(self, action_encoding_dim, num_res_blocks, num_channels, reward_head_channels, reward_hidden,
                 support_size, flat_reward, lstm_hidden_size):
        super().__init__()
        self.action_encoding_dim = action_encoding_dim
        self.flat_reward = flat_reward
        self.activation = nn.ReLU(inplace=True)
        self.conv = nn.Conv2d(num_channels, num_channels - action_encoding_dim, 3, 1, 1, bias=False)
        self.norm_common = nn.BatchNorm2d(num_channels - action_encoding_dim)
        self.resblocks = nn.ModuleList([ResBlock(num_channels - action_encoding_dim) for _ in range(num_res_blocks)])
        self.conv1x1_reward = nn.Conv2d(num_channels - action_encoding_dim, reward_head_channels, 1)
        self.norm_reward = nn.BatchNorm2d(reward_head_channels)
        self.lstm = nn.LSTM(input_size=flat_reward, hidden_size=lstm_hidden_size)
        self.norm_value_prefix = nn.BatchNorm1d(lstm_hidden_size)
        self.fc_reward_head = mlp(lstm_hidden_size, reward_hidden, support_size)

    def forward(self, state_action_encoding, reward_hidden_state):
        state_encoding = state_action_encoding[:, :-self.action_encoding_dim, :, :]
        x = self.norm_common(self.conv(state_action_encoding))
        x = x + state_encoding
        x = self.activation(x)
        for b in self.resblocks:
            x = b(x)
        next_latent_state = x
        x = self.activation(self.norm_reward(self.conv1x1_reward(next_latent_state)))
        x = x.reshape(-1, self.flat_reward).unsqueeze(0)
        value_prefix, next_reward_hidden_state = self.lstm(x, reward_hidden_state)
        value_prefix = self.activation(self.norm_value_prefix(value_prefix.squeeze(0)))
        value_prefix = self.fc_reward_head(value_prefix)
        return next_latent_state, next_reward_hidden_state, value_prefix


class EfficientZeroModel(nn.Module):  # efficientzero_model.py:20-382 (inference graph only)
    def __init__(self, observation_shape=(4, 96, 96), action_space_size=6, lstm_hidden_size=512, num_res_blocks=1,
                 num_channels=64, reward_head_channels=16, value_head_channels=16, policy_head_channels=16,
                 reward_head_hidden_channels=(32,), value_head_hidden_channels=(32,), policy_head_hidden_channels=(32,),
                 reward_support_range=(-300., 301., 1.), value_support_range=(-300., 301., 1.), downsample=True,
                 discrete_action_encoding_type='one_hot'):
        super().__init__()
        self.action_space_size = action_space_size
        assert discrete_action_encoding_type in ('one_hot', 'not_one_hot')
        self.discrete_action_encoding_type = discrete_action_encoding_type
        self.action_encoding_dim = action_space_size if discrete_action_encoding_type == 'one_hot' else 1   # efficientzero_model.py:103-108
        self.lstm_hidden_size = lstm_hidden_size
        self.reward_support_size = len(torch.arange(*reward_support_range))
        self.value_support_size = len(torch.arange(*value_support_range))
        if observation_shape[1] == 96:
            latent_size = math.ceil(observation_shape[1] / 16) * math.ceil(observation_shape[2] / 16)
        elif observation_shape[1] == 64:
            latent_size = math.ceil(observation_shape[1] / 8) * math.ceil(observation_shape[2] / 8)
        else:
            latent_size = observation_shape[1] * observation_shape[2]
        hw = latent_size if downsample else observation_shape[1] * observation_shape[2]
        self.representation_network = RepresentationNetwork(observation_shape, num_res_blocks, num_channels, downsample)
        self.dynamics_network = EZDynamicsNetwork(self.action_encoding_dim, num_res_blocks, num_channels + self.action_encoding_dim,
                                                  reward_head_channels, reward_head_hidden_channels[0],
                                                  self.reward_support_size, reward_head_channels * hw, lstm_hidden_size)
        self.prediction_network = PredictionNetwork(action_space_size, num_res_blocks, num_channels, value_head_channels,
                                                    policy_head_channels, value_head_hidden_channels[0],
                                                    policy_head_hidden_channels[0], self.value_support_size,
                                                    value_head_channels * hw, policy_head_channels * hw)

    def initial_inference(self, obs):
        batch_size = obs.size(0)
        latent_state = self.representation_network(obs)
        policy_logits, value = self.prediction_network(latent_state)
        reward_hidden_state = (torch.zeros(1, batch_size, self.lstm_hidden_size).to(obs.device),
                               torch.zeros(1, batch_size, self.lstm_hidden_size).to(obs.device))
        return EZNetworkOutput(value, [0. for _ in range(batch_size)], policy_logits, latent_state, reward_hidden_state)

    def recurrent_inference(self, latent_state, reward_hidden_state, action):
        action_encoding = _action_encoding(self, latent_state, action)
        state_action_encoding = torch.cat((latent_state, action_encoding), dim=1)
        next_latent_state, reward_hidden_state, value_prefix = self.dynamics_network(state_action_encoding,
                                                                                    reward_hidden_state)
        policy_logits, value = self.prediction_network(next_latent_state)
        return EZNetworkOutput(value, value_prefix, policy_logits, next_latent_state, reward_hidden_state)


def _swap_activation(mod, make):
    """every nn.ReLU of a module tree (attributes and nn.Sequential members) replaced by make()"""
    for name, child in mod.named_children():
        if isinstance(child, nn.ReLU):
            setattr(mod, name, make())
        else:
            _swap_activation(child, make)


class SampledPredictionNetwork(PredictionNetwork):  # sampled_efficientzero_model.py:490-660, discrete actions, norm_type='BN'
    """the same prediction network; the reference names the two head MLPs fc_value_head / fc_policy_head (:587-623)"""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.fc_value_head, self.fc_policy_head = self.fc_value, self.fc_policy
        del self.fc_value, self.fc_policy

    def forward(self, latent_state):
        for b in self.resblocks:
            latent_state = b(latent_state)
        value = self.activation(self.norm_value(self.conv1x1_value(latent_state)))
        policy = self.activation(self.norm_policy(self.conv1x1_policy(latent_state)))
        value = self.fc_value_head(value.reshape(-1, self.flat_value))
        policy = self.fc_policy_head(policy.reshape(-1, self.flat_policy))
        return policy, value


class SampledEfficientZeroModel(EfficientZeroModel):  # lzero/model/sampled_efficientzero_model.py:17-485, discrete actions, norm_type='BN'
    """With ``continuous_action_space=False`` the reference's conv Sampled EfficientZero has the EfficientZero network's layers (same tower, same
    dynamics network :353-445, MLP policy head over the action space :610-623) under its own defaults -- GELU(approximate='tanh') everywhere
    (:40) and 256-wide head MLPs (:29-31) -- and other names for the prediction heads.  Checked against the reference module itself in
    tests/test_torch_models_vs_reference.py."""

    def __init__(self, observation_shape=(4, 64, 64), action_space_size=6, num_of_sampled_actions=6, continuous_action_space=False,
                 norm_type='BN', activation=None, reward_head_hidden_channels=(256,), value_head_hidden_channels=(256,),
                 policy_head_hidden_channels=(256,), **kw):
        assert not continuous_action_space and norm_type == 'BN', "restated for discrete actions and BatchNorm (the reference's Atari configuration)"
        super().__init__(observation_shape=observation_shape, action_space_size=action_space_size,
                         reward_head_hidden_channels=reward_head_hidden_channels, value_head_hidden_channels=value_head_hidden_channels,
                         policy_head_hidden_channels=policy_head_hidden_channels, **kw)
        self.num_of_sampled_actions = num_of_sampled_actions
        self.continuous_action_space = False
        p = self.prediction_network
        self.prediction_network = SampledPredictionNetwork(
            action_space_size, len(p.resblocks), p.conv1x1_value.in_channels, p.conv1x1_value.out_channels, p.conv1x1_policy.out_channels,
            p.fc_value[0].out_features, p.fc_policy[0].out_features, p.fc_value[-1].out_features, p.flat_value, p.flat_policy)
        # sampled_efficientzero_model.py:177-218 hands `activation` to the dynamics network ONLY: the representation network keeps its default
        # ReLU (common.py:718) and the prediction network its default GELU (sampled_efficientzero_model.py:508)
        _swap_activation(self.prediction_network, lambda: nn.GELU(approximate='tanh'))
        if activation is None or isinstance(activation, nn.GELU) or (isinstance(activation, str) and activation.lower() == 'gelu'):
            _swap_activation(self.dynamics_network, lambda: nn.GELU(approximate='tanh'))


class MZDynamicsNetwork(nn.Module):  # lzero/model/muzero_model.py:419-538
    def __init__(self, action_encoding_dim, num_res_blocks, num_channels, reward_head_channels, reward_hidden,
                 support_size, flat_reward):
        super().__init__()
        self.action_encoding_dim = action_encoding_dim
        self.flat_reward = flat_reward
        self.activation = nn.ReLU(inplace=True)
        self.conv = nn.Conv2d(num_channels, num_channels - action_encoding_dim, 3, 1, 1, bias=False)
        self.norm_common = nn.BatchNorm2d(num_channels - action_encoding_dim)
        self.resblocks = nn.ModuleList([ResBlock(num_channels - action_encoding_dim) for _ in range(num_res_blocks)])
        self.conv1x1_reward = nn.Conv2d(num_channels - action_encoding_dim, reward_head_channels, 1)
        self.norm_reward = nn.BatchNorm2d(reward_head_channels)
        self.fc_reward_head = mlp(flat_reward, reward_hidden, support_size)

    def forward(self, state_action_encoding):
        state_encoding = state_action_encoding[:, :-self.action_encoding_dim, :, :]
        x = self.norm_common(self.conv(state_action_encoding))
        x = x + state_encoding
        x = self.activation(x)
        for b in self.resblocks:
            x = b(x)
        next_latent_state = x
        x = self.activation(self.norm_reward(self.conv1x1_reward(next_latent_state)))
        reward = self.fc_reward_head(x.view(x.shape[0], -1))
        return next_latent_state, reward


class MuZeroModel(nn.Module):  # lzero/model/muzero_model.py:20-374 (inference graph only)
    def __init__(self, observation_shape=(4, 96, 96), action_space_size=6, num_res_blocks=1, num_channels=64,
                 reward_head_channels=16, value_head_channels=16, policy_head_channels=16,
                 reward_head_hidden_channels=(32,), value_head_hidden_channels=(32,), policy_head_hidden_channels=(32,),
                 reward_support_range=(-300., 301., 1.), value_support_range=(-300., 301., 1.), downsample=True,
                 discrete_action_encoding_type='one_hot'):
        super().__init__()
        self.action_space_size = action_space_size
        assert discrete_action_encoding_type in ('one_hot', 'not_one_hot')
        self.discrete_action_encoding_type = discrete_action_encoding_type
        self.action_encoding_dim = action_space_size if discrete_action_encoding_type == 'one_hot' else 1   # muzero_model.py:105-110
        self.reward_support_size = len(torch.arange(*reward_support_range))
        self.value_support_size = len(torch.arange(*value_support_range))
        if observation_shape[1] == 96:
            latent_size = math.ceil(observation_shape[1] / 16) * math.ceil(observation_shape[2] / 16)
        elif observation_shape[1] == 64:
            latent_size = math.ceil(observation_shape[1] / 8) * math.ceil(observation_shape[2] / 8)
        else:
            latent_size = observation_shape[1] * observation_shape[2]
        hw = latent_size if downsample else observation_shape[1] * observation_shape[2]
        self.representation_network = RepresentationNetwork(observation_shape, num_res_blocks, num_channels, downsample)
        self.dynamics_network = MZDynamicsNetwork(self.action_encoding_dim, num_res_blocks, num_channels + self.action_encoding_dim,
                                                  reward_head_channels, reward_head_hidden_channels[0],
                                                  self.reward_support_size, reward_head_channels * hw)
        self.prediction_network = PredictionNetwork(action_space_size, num_res_blocks, num_channels, value_head_channels,
                                                    policy_head_channels, value_head_hidden_channels[0],
                                                    policy_head_hidden_channels[0], self.value_support_size,
                                                    value_head_channels * hw, policy_head_channels * hw)

    def initial_inference(self, obs):
        batch_size = obs.size(0)
        latent_state = self.representation_network(obs)
        policy_logits, value = self.prediction_network(latent_state)
        return MZNetworkOutput(value, [0. for _ in range(batch_size)], policy_logits, latent_state)

    def recurrent_inference(self, latent_state, action):
        action_encoding = _action_encoding(self, latent_state, action)
        next_latent_state, reward = self.dynamics_network(torch.cat((latent_state, action_encoding), dim=1))
        policy_logits, value = self.prediction_network(next_latent_state)
        return MZNetworkOutput(value, reward, policy_logits, next_latent_state)


# ------------------------------------------------------------------------------------------------
# vector-observation (MLP) model family
# ------------------------------------------------------------------------------------------------
def _norm1d(norm_type, n):
    """ding.torch_utils.build_normalization(norm_type, dim=1): 'BN' -> BatchNorm1d, 'LN' -> LayerNorm."""
    return {"BN": nn.BatchNorm1d, "LN": nn.LayerNorm}[norm_type](n)


def mlp_v2(cin, hidden, cout, activation, norm_type, output_activation, output_norm):
    """lzero/model/common.py:28-98 MLP_V2 (hidden is a list), and -- with hidden = [h] * (layer_num - 1) --
    ding.torch_utils.MLP(in, h, out, layer_num, ...) restated from DI-engine's published source: every layer is
    Linear [, norm] [, activation]; the last one takes norm / activation only when output_norm / output_activation."""
    dims = [cin] + list(hidden) + [cout]
    layers = []
    for i in range(len(dims) - 1):
        last = i == len(dims) - 2
        layers.append(nn.Linear(dims[i], dims[i + 1]))
        if norm_type and (not last or output_norm):
            layers.append(_norm1d(norm_type, dims[i + 1]))
        if activation is not None and (not last or output_activation):
            layers.append(activation)
    return nn.Sequential(*layers)


class RepresentationNetworkMLP(nn.Module):  # common.py:790-851
    def __init__(self, observation_shape, hidden_channels, activation=None, norm_type="BN"):
        super().__init__()
        activation = activation if activation is not None else nn.GELU(approximate="tanh")
        self.fc_representation = mlp_v2(observation_shape, [hidden_channels], hidden_channels, activation, norm_type, False, False)
        self.norm = nn.LayerNorm(hidden_channels)  # final_norm_option_in_encoder = 'LayerNorm'

    def forward(self, x):
        return self.norm(self.fc_representation(x.float()))


class PredictionNetworkMLP(nn.Module):  # common.py:1218-1295 (discrete policy head)
    def __init__(self, action_space_size, num_channels, value_hidden, policy_hidden, support, activation, norm_type):
        super().__init__()
        self.fc_prediction_common = mlp_v2(num_channels, [num_channels], num_channels, activation, norm_type, True, True)
        self.fc_value_head = mlp_v2(num_channels, value_hidden, support, activation, norm_type, False, False)
        self.fc_policy_head = mlp_v2(num_channels, policy_hidden, action_space_size, activation, norm_type, False, False)

    def forward(self, z):
        x = self.fc_prediction_common(z)
        return self.fc_policy_head(x), self.fc_value_head(x)


class ReparameterizationHead(nn.Module):
    """ding.model.common.ReparameterizationHead (un-vendored DI-engine; restated from its published source, PARITY
    UNPINNED): main = MLP(in, in, in, layer_num, activation, norm) ; mu = Linear ; sigma_type 'conditioned':
    sigma = exp(clamp(Linear(x), -20, 2)) ; 'fixed': constant ; bound_type 'tanh' squashes mu."""

    def __init__(self, input_size, output_size, layer_num=2, sigma_type="conditioned", fixed_sigma_value=0.3,
                 activation=None, norm_type=None, bound_type=None):
        super().__init__()
        self.sigma_type, self.bound_type, self.fixed_sigma_value = sigma_type, bound_type, fixed_sigma_value
        self.main = mlp_v2(input_size, [input_size] * (layer_num - 1), input_size, activation or nn.ReLU(), norm_type, True, True)
        self.mu = nn.Linear(input_size, output_size)
        if sigma_type == "conditioned":
            self.log_sigma_layer = nn.Linear(input_size, output_size)

    def forward(self, x):
        x = self.main(x)
        mu = self.mu(x)
        if self.bound_type == "tanh":
            mu = torch.tanh(mu)
        if self.sigma_type == "conditioned":
            sigma = torch.exp(torch.clamp(self.log_sigma_layer(x), -20, 2))
        else:
            sigma = torch.full_like(mu, self.fixed_sigma_value)
        return {"mu": mu, "sigma": sigma}


class SampledPredictionNetworkMLP(nn.Module):  # sampled_efficientzero_model_mlp.py: class PredictionNetworkMLP
    def __init__(self, continuous, action_space_size, num_channels, value_hidden, policy_hidden, support, activation,
                 norm_type, sigma_type, fixed_sigma_value, bound_type):
        super().__init__()
        self.continuous = continuous
        self.fc_prediction_common = mlp_v2(num_channels, [num_channels], num_channels, activation, norm_type, True, True)
        self.fc_value_head = mlp_v2(num_channels, [value_hidden[0]], support, activation, norm_type, False, False)
        if continuous:
            self.fc_policy_head = ReparameterizationHead(num_channels, action_space_size, 2, sigma_type, fixed_sigma_value,
                                                         nn.ReLU(), None, bound_type)
        else:
            self.fc_policy_head = mlp_v2(num_channels, [policy_hidden[0]], action_space_size, activation, norm_type, False, False)

    def forward(self, z):
        x = self.fc_prediction_common(z)
        value = self.fc_value_head(x)
        policy = self.fc_policy_head(x)
        if self.continuous:
            policy = torch.cat([policy["mu"], policy["sigma"]], dim=-1)
        return policy, value


class MZDynamicsNetworkMLP(nn.Module):  # muzero_model_mlp.py:340-442
    def __init__(self, action_encoding_dim, num_channels, reward_hidden, support, activation, norm_type, res):
        super().__init__()
        self.action_encoding_dim, self.res = action_encoding_dim, res
        L = num_channels - action_encoding_dim
        if res:
            self.fc_dynamics_1 = mlp_v2(num_channels, [L], L, activation, norm_type, True, True)
            self.fc_dynamics_2 = mlp_v2(L, [L], L, activation, norm_type, True, True)
        else:
            self.fc_dynamics = mlp_v2(num_channels, [L], L, activation, norm_type, True, True)
        self.fc_reward_head = mlp_v2(L, reward_hidden, support, activation, norm_type, False, False)

    def forward(self, sa):
        if self.res:
            nxt = self.fc_dynamics_1(sa) + sa[:, :-self.action_encoding_dim]
            enc = self.fc_dynamics_2(nxt)
        else:
            nxt = self.fc_dynamics(sa)
            enc = nxt
        return nxt, self.fc_reward_head(enc)


class EZDynamicsNetworkMLP(nn.Module):  # efficientzero_model_mlp.py: class DynamicsNetworkMLP
    def __init__(self, action_encoding_dim, num_channels, lstm_hidden_size, reward_hidden, support, activation, norm_type, res):
        super().__init__()
        self.action_encoding_dim, self.res = action_encoding_dim, res
        L = num_channels - action_encoding_dim
        if res:
            self.fc_dynamics_1 = mlp_v2(num_channels, [L], L, activation, norm_type, True, True)
            self.fc_dynamics_2 = mlp_v2(L, [L], L, activation, norm_type, True, True)
        else:
            self.fc_dynamics = mlp_v2(num_channels, [L], L, activation, norm_type, True, True)
        self.lstm = nn.LSTM(input_size=L, hidden_size=lstm_hidden_size)
        self.fc_reward_head = mlp_v2(lstm_hidden_size, [reward_hidden[0]], support, activation, norm_type, False, False)

    def forward(self, sa, reward_hidden_state):
        if self.res:
            nxt = self.fc_dynamics_1(sa) + sa[:, :-self.action_encoding_dim]
            enc = self.fc_dynamics_2(nxt)
        else:
            nxt = self.fc_dynamics(sa)
            enc = nxt
        vp, hc = self.lstm(enc.unsqueeze(0), reward_hidden_state)
        return nxt, hc, self.fc_reward_head(vp.squeeze(0))


def _encode_action(action, latent, continuous, action_space_size, encoding):
    """the action-encoding preamble of every MLP model's ``_dynamics`` (e.g. muzero_model_mlp.py:300-322)"""
    if continuous:
        if action.dim() == 1:
            action = action.unsqueeze(-1)
        elif action.dim() == 3:
            action = action.squeeze(-1)
        enc = action
    elif encoding == "one_hot":
        if action.dim() == 1:
            action = action.unsqueeze(-1)
        enc = torch.zeros(action.shape[0], action_space_size, device=action.device)
        enc.scatter_(1, action.long(), 1)
    else:
        enc = action / action_space_size
        if enc.dim() == 1:
            enc = enc.unsqueeze(-1)
    return torch.cat((latent, enc.to(latent.device).float()), dim=1)


def renormalize(x):  # lzero/model/utils.py:242-271 with first_dim = 1 on a [B, K] tensor
    mx, _ = torch.max(x, dim=1, keepdim=True)
    mn, _ = torch.min(x, dim=1, keepdim=True)
    den = mx - mn
    den[den < 1e-8] = 1e-8
    return (x - mn) / den


class MuZeroModelMLP(nn.Module):  # lzero/model/muzero_model_mlp.py:13-338 (inference graph only)
    def __init__(self, observation_shape=4, action_space_size=2, latent_state_dim=128, reward_head_hidden_channels=(32,),
                 value_head_hidden_channels=(32,), policy_head_hidden_channels=(32,), support_range=(-300., 301., 1.),
                 norm_type="BN", discrete_action_encoding_type="one_hot", res_connection_in_dynamics=False,
                 reward_support_range=None, value_support_range=None, categorical_distribution=True, state_norm=False):
        super().__init__()
        self.action_space_size, self.encoding = action_space_size, discrete_action_encoding_type
        self.state_norm = state_norm
        # muzero_model_mlp.py:72-77: the two heads are sized by their own supports; one output each without the categorical representation
        self.support_size = len(torch.arange(*(value_support_range or support_range))) if categorical_distribution else 1
        self.reward_support_size = len(torch.arange(*(reward_support_range or value_support_range or support_range))) if categorical_distribution else 1
        enc = action_space_size if discrete_action_encoding_type == "one_hot" else 1
        act = nn.ReLU(inplace=True)
        self.representation_network = RepresentationNetworkMLP(observation_shape, latent_state_dim, None, norm_type)
        self.dynamics_network = MZDynamicsNetworkMLP(enc, latent_state_dim + enc, list(reward_head_hidden_channels),
                                                     self.reward_support_size, act, norm_type, res_connection_in_dynamics)
        self.prediction_network = PredictionNetworkMLP(action_space_size, latent_state_dim, list(value_head_hidden_channels),
                                                       list(policy_head_hidden_channels), self.support_size, act, norm_type)

    def initial_inference(self, obs):
        z = self.representation_network(obs)
        if self.state_norm:   # muzero_model_mlp.py:220-221
            z = renormalize(z)
        policy_logits, value = self.prediction_network(z)
        return MZNetworkOutput(value, [0. for _ in range(obs.size(0))], policy_logits, z)

    def recurrent_inference(self, latent_state, action):
        sa = _encode_action(action, latent_state, False, self.action_space_size, self.encoding)
        nxt, reward = self.dynamics_network(sa)
        if self.state_norm:   # muzero_model_mlp.py:291-295 (the reward was computed on the un-normalised next latent)
            nxt = renormalize(nxt)
        policy_logits, value = self.prediction_network(nxt)
        return MZNetworkOutput(value, reward, policy_logits, nxt)


class EfficientZeroModelMLP(nn.Module):  # lzero/model/efficientzero_model_mlp.py (inference graph only)
    def __init__(self, observation_shape=4, action_space_size=2, lstm_hidden_size=512, latent_state_dim=256,
                 reward_head_hidden_channels=(32,), value_head_hidden_channels=(32,), policy_head_hidden_channels=(32,),
                 support_range=(-300., 301., 1.), norm_type="BN", discrete_action_encoding_type="one_hot",
                 res_connection_in_dynamics=False, categorical_distribution=True, state_norm=False):
        super().__init__()
        self.action_space_size, self.encoding, self.lstm_hidden_size = action_space_size, discrete_action_encoding_type, lstm_hidden_size
        self.state_norm = state_norm
        self.support_size = len(torch.arange(*support_range)) if categorical_distribution else 1   # efficientzero_model_mlp.py:76-81
        enc = action_space_size if discrete_action_encoding_type == "one_hot" else 1
        act = nn.ReLU(inplace=True)
        self.representation_network = RepresentationNetworkMLP(observation_shape, latent_state_dim, None, norm_type)
        self.dynamics_network = EZDynamicsNetworkMLP(enc, latent_state_dim + enc, lstm_hidden_size, list(reward_head_hidden_channels),
                                                     self.support_size, act, norm_type, res_connection_in_dynamics)
        self.prediction_network = PredictionNetworkMLP(action_space_size, latent_state_dim, list(value_head_hidden_channels),
                                                       list(policy_head_hidden_channels), self.support_size, act, norm_type)

    def initial_inference(self, obs):
        z = self.representation_network(obs)
        if self.state_norm:   # efficientzero_model_mlp.py:232-233
            z = renormalize(z)
        policy_logits, value = self.prediction_network(z)
        B = obs.size(0)
        hc = (torch.zeros(1, B, self.lstm_hidden_size).to(obs.device), torch.zeros(1, B, self.lstm_hidden_size).to(obs.device))
        return EZNetworkOutput(value, [0. for _ in range(B)], policy_logits, z, hc)

    def recurrent_inference(self, latent_state, reward_hidden_state, action):
        sa = _encode_action(action, latent_state, False, self.action_space_size, self.encoding)
        nxt, hc, vp = self.dynamics_network(sa, reward_hidden_state)
        if self.state_norm:   # efficientzero_model_mlp.py:307-308
            nxt = renormalize(nxt)
        policy_logits, value = self.prediction_network(nxt)
        return EZNetworkOutput(value, vp, policy_logits, nxt, hc)


class SampledEfficientZeroModelMLP(nn.Module):  # lzero/model/sampled_efficientzero_model_mlp.py (inference graph only)
    def __init__(self, observation_shape=5, action_space_size=1, latent_state_dim=256, lstm_hidden_size=512,
                 reward_head_hidden_channels=(256,), value_head_hidden_channels=(256,), policy_head_hidden_channels=(256,),
                 support_range=(-300., 301., 1.), continuous_action_space=True, num_of_sampled_actions=20,
                 sigma_type="conditioned", fixed_sigma_value=0.3, bound_type=None, norm_type="LN",
                 discrete_action_encoding_type="one_hot", res_connection_in_dynamics=True, categorical_distribution=True, state_norm=False):
        super().__init__()
        self.action_space_size, self.encoding, self.lstm_hidden_size = action_space_size, discrete_action_encoding_type, lstm_hidden_size
        self.continuous_action_space, self.num_of_sampled_actions = continuous_action_space, num_of_sampled_actions
        self.state_norm = state_norm
        self.support_size = len(torch.arange(*support_range)) if categorical_distribution else 1   # sampled_efficientzero_model_mlp.py:95-100
        enc = action_space_size if (continuous_action_space or discrete_action_encoding_type == "one_hot") else 1
        act = nn.GELU(approximate="tanh")
        self.representation_network = RepresentationNetworkMLP(observation_shape, latent_state_dim, act, norm_type)
        self.dynamics_network = EZDynamicsNetworkMLP(enc, latent_state_dim + enc, lstm_hidden_size, list(reward_head_hidden_channels),
                                                     self.support_size, act, norm_type, res_connection_in_dynamics)
        self.prediction_network = SampledPredictionNetworkMLP(continuous_action_space, action_space_size, latent_state_dim,
                                                              list(value_head_hidden_channels), list(policy_head_hidden_channels),
                                                              self.support_size, act, norm_type, sigma_type, fixed_sigma_value, bound_type)

    def initial_inference(self, obs):
        z = self.representation_network(obs)
        if self.state_norm:   # sampled_efficientzero_model_mlp.py:269-270
            z = renormalize(z)
        policy_logits, value = self.prediction_network(z)
        B = obs.size(0)
        hc = (torch.zeros(1, B, self.lstm_hidden_size).to(obs.device), torch.zeros(1, B, self.lstm_hidden_size).to(obs.device))
        return EZNetworkOutput(value, [0. for _ in range(B)], policy_logits, z, hc)

    def recurrent_inference(self, latent_state, reward_hidden_state, action):
        sa = _encode_action(action, latent_state, self.continuous_action_space, self.action_space_size, self.encoding)
        nxt, hc, vp = self.dynamics_network(sa, reward_hidden_state)
        if self.state_norm:   # sampled_efficientzero_model_mlp.py:356-360
            nxt = renormalize(nxt)
        policy_logits, value = self.prediction_network(nxt)
        return EZNetworkOutput(value, vp, policy_logits, nxt, hc)



class InverseScalarTransform(object):  # scaling_transform.py:64-92
    def __init__(self, support_range=(-300., 301., 1.), categorical_distribution=True, device="cpu"):
        self.value_support = torch.arange(*support_range, dtype=torch.float32).unsqueeze(0).to(device)
        self.categorical_distribution = categorical_distribution

    def __call__(self, logits, epsilon=0.001):
        if self.categorical_distribution:
            value_probs = torch.softmax(logits, dim=1)
            value = value_probs.mul_(self.value_support).sum(1, keepdim=True)
        else:
            value = logits
        tmp = ((torch.sqrt(1 + 4 * epsilon * (torch.abs(value) + 1 + epsilon)) - 1) / (2 * epsilon))
        return torch.sign(value) * (tmp * tmp - 1)


def synthetic_init(model, seed=0):
    """Seeded synthetic weights: default torch init, BN running stats randomised,
    zero-initialised last linear layers re-drawn N(0, 0.05) so that logits are not all zero."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() >= 2:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / math.sqrt(fan_in)))
            elif name.endswith("bias") or ".bias_" in name:  # nn.LSTM names its biases bias_ih_l0 / bias_hh_l0
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
        for m in model.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_(1.0 + 0.1 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
            if isinstance(m, nn.LayerNorm):
                m.weight.copy_(1.0 + 0.1 * torch.randn(m.weight.shape, generator=g))
        for head in ("fc_value", "fc_policy", "fc_reward_head"):
            for m in model.modules():
                if hasattr(m, head):
                    last = getattr(m, head)[-1]
                    last.weight.copy_(torch.randn(last.weight.shape, generator=g) * 0.05)
                    last.bias.copy_(torch.randn(last.bias.shape, generator=g) * 0.05)
    model.eval()
    return model
