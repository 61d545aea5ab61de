from .muzero_collector import MuZeroVectorCollector  # noqa: F401
