"""Drop-in for lzero/mcts/ctree/ctree_muzero/mz_tree.pyx (``batch_traverse`` takes
``deterministic=False`` like mz_tree.pyx:95-98)."""
from .._tree_common import make_module as _make

globals().update(_make(1, has_deterministic_flag=True))
