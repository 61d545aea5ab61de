"""Drop-in for lzero/mcts/ctree/ctree_efficientzero/ez_tree.pyx: ``Roots``, ``MinMaxStatsList``,
``ResultsWrapper``, ``batch_traverse``, ``batch_backpropagate`` -- trees in HBM, HIP kernels."""
from .._tree_common import make_module as _make

globals().update(_make(0, has_deterministic_flag=False))
