"""ding.torch_utils: ResBlock and MLP as the reference's model files use them (restated, see ../../README.md)."""
import torch.nn as nn

from .network.normalization import build_normalization


def _conv_block(cin, cout, stride, activation, norm_type, bias):
    # DI-engine's conv2d_block: Sequential(conv3x3 pad 1, [norm], [activation])
    mods = [nn.Conv2d(cin, cout, 3, stride, 1, bias=bias)]
    if norm_type is not None:
        mods.append(build_normalization(norm_type, dim=2)(cout))
    if activation is not None:
        mods.append(activation)
    return nn.Sequential(*mods)


class ResBlock(nn.Module):
    """res_type 'basic': x -> conv-norm-act -> conv-norm -> (+ x) -> act
       res_type 'downsample': both the main path's first conv and the identity path (conv3, no norm, no act) have stride 2"""

    def __init__(self, in_channels, activation=nn.ReLU(), norm_type='BN', res_type='basic', bias=True, out_channels=None):
        super().__init__()
        assert res_type in ('basic', 'downsample'), res_type
        out_channels = in_channels if out_channels is None else out_channels
        self.act = activation
        self.res_type = res_type
        self.conv1 = _conv_block(in_channels, out_channels, 2 if res_type == 'downsample' else 1, self.act, norm_type, bias)
        self.conv2 = _conv_block(out_channels, out_channels, 1, None, norm_type, bias)
        if res_type == 'downsample':
            self.conv3 = _conv_block(in_channels, out_channels, 2, None, None, bias)

    def forward(self, x):
        y = self.conv2(self.conv1(x))
        shortcut = self.conv3(x) if self.res_type == 'downsample' else x
        return self.act(y + shortcut)


def MLP(in_channels, hidden_channels, out_channels, layer_num, layer_fn=None, activation=None, norm_type=None,
        use_dropout=False, dropout_probability=0.5, output_activation=True, output_norm=True,
        last_linear_layer_init_zero=False):
    """Sequential of layer_num Linear layers; every hidden one is followed by [norm] [activation], the last one only when
    output_norm / output_activation ask for it; optionally the last Linear starts at zero."""
    assert layer_num >= 1 and not use_dropout and layer_fn is None
    widths = [in_channels] + [hidden_channels] * (layer_num - 1) + [out_channels]
    mods = []
    for i in range(layer_num):
        final = i == layer_num - 1
        mods.append(nn.Linear(widths[i], widths[i + 1]))
        if norm_type is not None and (not final or output_norm):
            mods.append(build_normalization(norm_type, dim=1)(widths[i + 1]))
        if activation is not None and (not final or output_activation):
            mods.append(activation)
    if last_linear_layer_init_zero:
        for m in reversed(mods):
            if isinstance(m, nn.Linear):
                nn.init.zeros_(m.weight)
                nn.init.zeros_(m.bias)
                break
    return nn.Sequential(*mods)
