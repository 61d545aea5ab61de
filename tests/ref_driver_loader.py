"""TEST INFRASTRUCTURE: imports the reference's OWN search drivers -- /root/reference/lzero/mcts/tree_search/mcts_ctree.py, as it
lies -- in a container without DI-engine / easydict and without the reference's build system:

  lzero.mcts.ctree.ctree_{efficientzero,muzero,gumbel_muzero}  ->  the reference's own ctree sources compiled by oracle/build_ref.py
                                                                   (oracle/_ref/det: rand() -> 0 so that ties resolve reproducibly)
  lzero.policy                                                 ->  DiscreteSupport / InverseScalarTransform from the reference's real
                                                                   lzero/policy/scaling_transform.py (imported by path) and
                                                                   to_detach_cpu_numpy (lzero/policy/utils.py:746-763, four lines restated:
                                                                   utils.py itself imports DI-engine)
  easydict                                                     ->  tests/ref_stubs/easydict

The stubs live in sys.modules only while the driver module is being executed."""
import importlib.util
import os
import sys
import types

REF = "/root/reference/lzero"
_HERE = os.path.dirname(os.path.abspath(__file__))
_cache = {}


def available():
    return os.path.isfile(os.path.join(REF, "mcts", "tree_search", "mcts_ctree.py"))


def _to_detach_cpu_numpy(data_list):
    import torch
    if isinstance(data_list, torch.Tensor):
        return data_list.detach().cpu().numpy()
    if isinstance(data_list, list) and all(isinstance(d, torch.Tensor) for d in data_list):
        return [d.detach().cpu().numpy() for d in data_list]
    raise TypeError("The type of input must be torch.Tensor or List[torch.Tensor]")


def load():
    """-> namespace(driver = the reference's mcts_ctree module, ez_tree, mz_tree, gmz_tree = the compiled reference trees it uses);
    None when /root/reference (or its compiled ctree) is absent."""
    if "ns" in _cache:
        return _cache["ns"]
    _cache["ns"] = None
    if not available():
        return None
    from oracle import build_ref
    if not build_ref.build():
        return None
    ez_tree, mz_tree = build_ref.load("det")
    gmz_tree = build_ref.load_gumbel("det") or build_ref.load_gumbel()
    import ref_loader
    ref = ref_loader.load()
    stubs = os.path.join(_HERE, "ref_stubs")
    if stubs not in sys.path:
        sys.path.insert(0, stubs)
    names = {}

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        names[name] = m
        return m
    mod("lzero"); mod("lzero.mcts"); mod("lzero.mcts.ctree")
    mod("lzero.mcts.ctree.ctree_efficientzero", ez_tree=ez_tree)
    mod("lzero.mcts.ctree.ctree_muzero", mz_tree=mz_tree)
    mod("lzero.mcts.ctree.ctree_gumbel_muzero", gmz_tree=gmz_tree)
    mod("lzero.policy", DiscreteSupport=ref.scaling_transform.DiscreteSupport,
        InverseScalarTransform=ref.scaling_transform.InverseScalarTransform, to_detach_cpu_numpy=_to_detach_cpu_numpy)
    saved = {k: sys.modules.get(k) for k in names}
    sys.modules.update(names)
    try:
        spec = importlib.util.spec_from_file_location("lzref_mcts_ctree", os.path.join(REF, "mcts", "tree_search", "mcts_ctree.py"))
        driver = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(driver)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    _cache["ns"] = types.SimpleNamespace(driver=driver, ez_tree=ez_tree, mz_tree=mz_tree, gmz_tree=gmz_tree)
    return _cache["ns"]


class Mutable(object):
    """the reference driver assigns to the fields of the network output (mcts_ctree.py:837-848); the reference's own output type
    is a dataclass, the oracle restatement returns a namedtuple"""

    def __init__(self, nt):
        for k, v in nt._asdict().items():
            setattr(self, k, v)


class MutableOutputModel(object):
    def __init__(self, model, record=None):
        self._m = model
        self.record = record   # optional list: the raw network outputs of every call

    def eval(self):
        self._m.eval()
        return self

    def initial_inference(self, *a, **k):
        return Mutable(self._m.initial_inference(*a, **k))

    def recurrent_inference(self, *a, **k):
        out = self._m.recurrent_inference(*a, **k)
        if self.record is not None:
            self.record.append((a, out))
        return Mutable(out)


def driver_cfg(num_simulations, discount_factor=0.997, env_type="not_board_games", support=(-300., 301., 1.), **extra):
    from easydict import EasyDict
    return EasyDict(dict(num_simulations=num_simulations, discount_factor=discount_factor, lstm_horizon_len=5, device="cpu", env_type=env_type,
                         pb_c_base=19652, pb_c_init=1.25, value_delta_max=0.01, root_noise_weight=0.25,
                         model=dict(value_support_range=support, reward_support_range=support, categorical_distribution=True), **extra))
