"""ding.model.common.ReparameterizationHead as sampled_efficientzero_model_mlp.py:482-491 uses it (restated, see ../../README.md)"""
import torch
import torch.nn as nn

from ding.torch_utils import MLP


class ReparameterizationHead(nn.Module):
    def __init__(self, input_size, output_size, layer_num=2, sigma_type=None, fixed_sigma_value=1.0, activation=nn.ReLU(),
                 norm_type=None, bound_type=None, hidden_size=None):
        super().__init__()
        assert sigma_type in ('fixed', 'independent', 'conditioned'), sigma_type
        assert bound_type in (None, 'tanh'), bound_type
        hidden_size = input_size if hidden_size is None else hidden_size
        self.sigma_type, self.bound_type, self.fixed_sigma_value = sigma_type, bound_type, fixed_sigma_value
        self.main = MLP(input_size, hidden_size, hidden_size, layer_num, activation=activation, norm_type=norm_type)
        self.mu = nn.Linear(hidden_size, output_size)
        if sigma_type == 'independent':
            self.log_sigma_param = nn.Parameter(torch.zeros(1, output_size))
        elif sigma_type == 'conditioned':
            self.log_sigma_layer = nn.Linear(hidden_size, output_size)

    def forward(self, x):
        x = self.main(x)
        mu = self.mu(x)
        if self.bound_type == 'tanh':
            mu = torch.tanh(mu)
        if self.sigma_type == 'fixed':
            sigma = torch.full_like(mu, self.fixed_sigma_value)
        elif self.sigma_type == 'independent':
            sigma = torch.exp(self.log_sigma_param + torch.zeros_like(mu))
        else:
            sigma = torch.exp(torch.clamp(self.log_sigma_layer(x), -20, 2))
        return {'mu': mu, 'sigma': sigma}
