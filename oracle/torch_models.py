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

NN parity is unpinned by the reference's own tests (they assert shapes only,
lzero/model/tests/test_efficientzero_model.py:89-127); the pin used here is this restatement on
shared random weights, plus the h^-1 known-answer of lzero/policy/tests/test_scaling_transform.py.
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
                 reward_support_range=(-300., 301., 1.), value_support_range=(-300., 301., 1.), downsample=True):
        super().__init__()
        self.action_space_size = action_space_size
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
        self.dynamics_network = EZDynamicsNetwork(action_space_size, num_res_blocks, num_channels + action_space_size,
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
        if len(action.shape) == 1:
            action = action.unsqueeze(-1)
        action_one_hot = torch.zeros(action.shape[0], self.action_space_size, device=action.device)
        action_one_hot.scatter_(1, action.long(), 1)
        action_encoding = action_one_hot.unsqueeze(-1).unsqueeze(-1).expand(
            latent_state.shape[0], self.action_space_size, latent_state.shape[2], latent_state.shape[3])
        state_action_encoding = torch.cat((latent_state, action_encoding), dim=1)
        next_latent_state, reward_hidden_state, value_prefix = self.dynamics_network(state_action_encoding,
                                                                                    reward_hidden_state)
        policy_logits, value = self.prediction_network(next_latent_state)
        return EZNetworkOutput(value, value_prefix, policy_logits, next_latent_state, reward_hidden_state)


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
                 reward_support_range=(-300., 301., 1.), value_support_range=(-300., 301., 1.), downsample=True):
        super().__init__()
        self.action_space_size = action_space_size
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
        self.dynamics_network = MZDynamicsNetwork(action_space_size, num_res_blocks, num_channels + action_space_size,
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
        if len(action.shape) == 1:
            action = action.unsqueeze(-1)
        action_one_hot = torch.zeros(action.shape[0], self.action_space_size, device=action.device)
        action_one_hot.scatter_(1, action.long(), 1)
        action_encoding = action_one_hot.unsqueeze(-1).unsqueeze(-1).expand(
            latent_state.shape[0], self.action_space_size, latent_state.shape[2], latent_state.shape[3])
        next_latent_state, reward = self.dynamics_network(torch.cat((latent_state, action_encoding), dim=1))
        policy_logits, value = self.prediction_network(next_latent_state)
        return MZNetworkOutput(value, reward, policy_logits, next_latent_state)


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
            elif name.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
        for m in model.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_(1.0 + 0.1 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
        for head in ("fc_value", "fc_policy", "fc_reward_head"):
            for m in model.modules():
                if hasattr(m, head):
                    last = getattr(m, head)[-1]
                    last.weight.copy_(torch.randn(last.weight.shape, generator=g) * 0.05)
                    last.bias.copy_(torch.randn(last.bias.shape, generator=g) * 0.05)
    model.eval()
    return model
