"""Seeded synthetic weights for the convolutional EfficientZero model (BASELINE configs[1]) in the reference's ``state_dict``
format -- the benchmark has no checkpoint to load (no network).  Tensor names and shapes follow
lzero/model/efficientzero_model.py (RepresentationNetwork + DownSample common.py:266-365,706-787, DynamicsNetwork :427-569,
PredictionNetwork common.py:1081-1216, DI-engine ResBlock / MLP sequentials).  Recipe: weights
N(0, 1/sqrt(fan_in)), biases N(0, 0.05), BatchNorm running statistics randomised (mean N(0, 0.1), var U(0.5, 1.5), affine
1 + 0.1 N / 0.1 N) -- in particular the zero-initialised last linear layers are NOT zero, otherwise every logit ties."""
import numpy as np


def efficientzero_state_dict(seed=0, observation_channels=4, action_space_size=6, num_channels=64, lstm_hidden_size=512,
                             head_channels=16, head_hidden=32, support_size=601, latent_pixels=36, muzero=False, downsample=True):
    """muzero=True: lzero/model/muzero_model.py's layout instead (no LSTM / norm_value_prefix; the reward MLP reads the
    flattened conv1x1 features, muzero_model.py:505-538).  downsample=False: the board-game representation network
    (conv3x3 + norm on the raw planes, common.py:735-741) with latent_pixels = H * W of the board."""
    rng = np.random.default_rng(seed)
    sd = {}
    C, C2, A, H, HC, HID, SUP, HW = num_channels, num_channels // 2, action_space_size, lstm_hidden_size, head_channels, head_hidden, support_size, latent_pixels

    def w(name, *shape):
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        sd[name] = (rng.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)

    def b(name, n):
        sd[name] = (0.05 * rng.standard_normal(n)).astype(np.float32)

    def bn(prefix, n):
        sd[prefix + ".weight"] = (1.0 + 0.1 * rng.standard_normal(n)).astype(np.float32)
        sd[prefix + ".bias"] = (0.1 * rng.standard_normal(n)).astype(np.float32)
        sd[prefix + ".running_mean"] = (0.1 * rng.standard_normal(n)).astype(np.float32)
        sd[prefix + ".running_var"] = (0.5 + rng.random(n)).astype(np.float32)

    def resblock(prefix, cin, cout, downsample=False):
        w(prefix + ".conv1.0.weight", cout, cin, 3, 3); bn(prefix + ".conv1.1", cout)
        w(prefix + ".conv2.0.weight", cout, cout, 3, 3); bn(prefix + ".conv2.1", cout)
        if downsample:
            w(prefix + ".conv3.0.weight", cout, cin, 3, 3)

    def mlp(prefix, cin, cout):
        w(prefix + ".0.weight", HID, cin); b(prefix + ".0.bias", HID); bn(prefix + ".1", HID)
        sd[prefix + ".3.weight"] = (0.05 * rng.standard_normal((cout, HID))).astype(np.float32)
        sd[prefix + ".3.bias"] = (0.05 * rng.standard_normal(cout)).astype(np.float32)

    if downsample:
        d = "representation_network.downsample_net."
        w(d + "conv1.weight", C2, observation_channels, 3, 3); bn(d + "norm1", C2)
        resblock(d + "resblocks1.0", C2, C2)
        resblock(d + "downsample_block", C2, C, downsample=True)
        resblock(d + "resblocks2.0", C, C)
        resblock(d + "resblocks3.0", C, C)
    else:
        w("representation_network.conv.weight", C, observation_channels, 3, 3); bn("representation_network.norm", C)
    resblock("representation_network.resblocks.0", C, C)
    d = "dynamics_network."
    w(d + "conv.weight", C, C + A, 3, 3); bn(d + "norm_common", C)
    resblock(d + "resblocks.0", C, C)
    w(d + "conv1x1_reward.weight", HC, C, 1, 1); b(d + "conv1x1_reward.bias", HC); bn(d + "norm_reward", HC)
    if muzero:
        mlp(d + "fc_reward_head", HC * HW, SUP)
    else:
        w(d + "lstm.weight_ih_l0", 4 * H, HC * HW); w(d + "lstm.weight_hh_l0", 4 * H, H)
        b(d + "lstm.bias_ih_l0", 4 * H); b(d + "lstm.bias_hh_l0", 4 * H)
        bn(d + "norm_value_prefix", H)
        mlp(d + "fc_reward_head", H, SUP)
    d = "prediction_network."
    resblock(d + "resblocks.0", C, C)
    w(d + "conv1x1_value.weight", HC, C, 1, 1); b(d + "conv1x1_value.bias", HC)
    w(d + "conv1x1_policy.weight", HC, C, 1, 1); b(d + "conv1x1_policy.bias", HC)
    bn(d + "norm_value", HC); bn(d + "norm_policy", HC)
    mlp(d + "fc_value", HC * HW, SUP)
    mlp(d + "fc_policy", HC * HW, A)
    return sd


def sharpen_state_dict(sd, policy_scale=1.0, value_scale=None):
    """A copy of ``sd`` whose policy head's last layer (``prediction_network.fc_policy.3``: weight and bias) is multiplied by
    ``policy_scale`` and whose value head's last layer by ``value_scale`` (default: the same factor).  The section-8d recipe draws those
    layers at N(0, 0.05): root priors come out near-uniform (max-prob ~0.24 of 6 actions) and 50 simulations build trees of depth 2-3.
    A trained agent's prior is sharp; x10 gives max-prob ~0.9 and search paths of depth 4-10 (bench.py's ``depth_sweep`` arms and the
    sharp-prior parity cases use it).  Works on numpy or torch values."""
    value_scale = policy_scale if value_scale is None else value_scale
    out = dict(sd)
    for head, k in (("prediction_network.fc_policy.3", policy_scale), ("prediction_network.fc_value.3", value_scale)):
        for part in (".weight", ".bias"):
            out[head + part] = sd[head + part] * k
    return out


def muzero_state_dict(seed=0, **kw):
    return efficientzero_state_dict(seed=seed, muzero=True, **kw)


def mlp_state_dict(spec, seed=0):
    """Seeded synthetic weights for a vector-observation model whose tensor names / shapes are listed in
    synthetic_mlp_specs.json (``spec`` = "muzero_mlp_cartpole" | "sampled_efficientzero_mlp_dmc"; generated by
    tests/golden/make_mlp_weight_specs.py).  Same recipe as above, decided by the tensor name: ``running_var`` U(0.5, 1.5),
    ``running_mean`` N(0, 0.1), 1-D ``weight`` (norm affine) 1 + 0.1 N, ``bias`` N(0, 0.05)..0.1, matrices N(0, 1/sqrt(fan_in))."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "synthetic_mlp_specs.json")) as f:
        shapes = json.load(f)[spec]
    rng = np.random.default_rng(seed)
    sd = {}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        if name.endswith("running_var"):
            v = 0.5 + rng.random(shape)
        elif name.endswith("running_mean"):
            v = 0.1 * rng.standard_normal(shape)
        elif name.endswith("bias"):
            v = 0.05 * rng.standard_normal(shape)
        elif len(shape) == 1:
            v = 1.0 + 0.1 * rng.standard_normal(shape)
        else:
            v = rng.standard_normal(shape) / np.sqrt(int(np.prod(shape[1:])))
        sd[name] = v.astype(np.float32)
    return sd
