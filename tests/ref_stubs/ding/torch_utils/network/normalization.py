import torch.nn as nn


def build_normalization(norm_type, dim=None):
    """'BN' -> BatchNorm{dim}d, 'LN' -> LayerNorm (the two the hot-path models use)"""
    if norm_type == 'BN':
        return {1: nn.BatchNorm1d, 2: nn.BatchNorm2d}[dim]
    if norm_type == 'LN':
        return nn.LayerNorm
    raise KeyError(norm_type)
