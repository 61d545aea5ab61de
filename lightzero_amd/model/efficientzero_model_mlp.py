"""``EfficientZeroModelMLP``  lzero/model/efficientzero_model_mlp.py (inference graph): MLP dynamics + value-prefix LSTM.
Pair it with ez_tree.Roots.  See muzero_model_mlp.py for the engine mechanics."""
from .muzero_model_mlp import _EngineModelMLP


class EfficientZeroModelMLP(_EngineModelMLP):
    _model_type = 3
    _uses_lstm = True
