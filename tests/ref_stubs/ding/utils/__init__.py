"""ding.utils: inert registry / aliases / rank helpers"""
from typing import List, Tuple, Union

SequenceType = Union[List, Tuple]


class _Registry(dict):
    def register(self, name, *a, **k):
        def deco(cls):
            self[name] = cls
            return cls
        return deco


MODEL_REGISTRY = _Registry()
POLICY_REGISTRY = _Registry()


def get_rank():
    return 0


def get_world_size():
    return 1


def set_pkg_seed(seed, use_cuda=True):
    import random
    import numpy as np
    import torch
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
