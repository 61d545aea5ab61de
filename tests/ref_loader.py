"""TEST INFRASTRUCTURE: imports the reference's OWN model / policy modules from /root/reference (read-only, this container
only) under a private package name, with tests/ref_stubs standing in for DI-engine (`ding`) and `ditk`.

    ref = load()            # None when /root/reference is absent (the GPU box)
    ref.common, ref.efficientzero_model, ref.muzero_model, ref.muzero_model_mlp, ref.efficientzero_model_mlp,
    ref.sampled_efficientzero_model_mlp, ref.sampled_efficientzero_model, ref.scaling_transform, ref.game_segment (lzero/mcts/buffer/game_segment.py)
"""
import importlib
import importlib.util
import os
import sys
import types

REF = "/root/reference/lzero"
_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_stubs")
_cache = {}


def available():
    return os.path.isdir(os.path.join(REF, "model"))


def load():
    if "ref" in _cache:
        return _cache["ref"]
    if not available():
        _cache["ref"] = None
        return None
    if _STUBS not in sys.path:
        sys.path.insert(0, _STUBS)
    if "transformers" not in sys.modules:
        # common.py:23 imports AutoModelForCausalLM / AutoTokenizer for its language encoders; nothing on this path uses them
        # and the real package takes ~20 s to import
        t = types.ModuleType("transformers")
        t.AutoModelForCausalLM = t.AutoTokenizer = t.AutoModel = object
        sys.modules["transformers"] = t
        _cache["fake_transformers"] = True
    # a package whose __path__ is the reference's model directory, WITHOUT running its __init__.py (which pulls in image
    # transforms etc.); the relative imports of the model files (`from .common import ...`) resolve inside it
    pkg = types.ModuleType("lzref_model")
    pkg.__path__ = [os.path.join(REF, "model")]
    sys.modules["lzref_model"] = pkg
    ns = types.SimpleNamespace()
    for name in ("utils", "common", "efficientzero_model", "muzero_model", "muzero_model_mlp", "efficientzero_model_mlp",
                 "sampled_efficientzero_model_mlp", "sampled_efficientzero_model"):
        setattr(ns, name, importlib.import_module("lzref_model." + name))
    spec = importlib.util.spec_from_file_location("lzref_scaling_transform", os.path.join(REF, "policy", "scaling_transform.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ns.scaling_transform = mod
    spec = importlib.util.spec_from_file_location("lzref_game_segment", os.path.join(REF, "mcts", "buffer", "game_segment.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ns.game_segment = mod
    if _cache.get("fake_transformers"):
        del sys.modules["transformers"]
    _cache["ref"] = ns
    return ns
