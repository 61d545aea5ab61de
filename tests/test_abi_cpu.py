"""CPU: liblz_mi355.so builds for gfx950, loads, exports every symbol include/lz_mi355.h declares,
and fails loudly (no fallback) when no GPU is visible."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "lz_mi355.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lz_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    from lightzero_amd import build
    path = build.build()
    lib = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 15
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from lightzero_amd import _lib as L
    assert L.lib().lz_device_count() == 0
    with pytest.raises(L.LzError):
        L.default_engine(0)
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    roots = ez_tree.Roots(1, [[0, 1]])
    with pytest.raises(L.LzError):
        roots.prepare_no_noise([0.0], [[0.0, 0.0]], [-1])


def test_synthetic_state_dict_matches_reference_layout():
    """the benchmark's seeded weights (product code, no oracle import) carry exactly the reference model's tensor names / shapes"""
    from oracle import torch_models as tm
    from lightzero_amd.model.synthetic import efficientzero_state_dict
    sd = efficientzero_state_dict(seed=0, action_space_size=6)
    ref = {k: tuple(v.shape) for k, v in tm.EfficientZeroModel(action_space_size=6).state_dict().items() if "num_batches" not in k}
    assert {k: tuple(v.shape) for k, v in sd.items()} == ref


def test_release_library_has_no_skip_work_knobs():
    """the switches that drop kernels / layers from a search (timing experiments) are compiled only into
    liblz_mi355_dbg.so (-DLZ_DEBUG_KNOBS); the release library must not even contain their names."""
    from lightzero_amd import build
    blob = open(build.build(), "rb").read()
    for knob in (b"LZ_DEBUG_SKIP", b"LZ_DEBUG_CHAIN_LAYERS", b"LZ_DEBUG_CHAIN_TS", b"LZ_DEBUG_LSTM_ROWS", b"LZ_DEBUG_LSTM_HOTW", b"LZ_DEBUG_HEADS_TS", b"LZ_DEBUG_TREE_TS"):
        assert knob not in blob, knob
    assert not hasattr(ctypes.CDLL(build.LIB), "lz_debug_read_chain_ts")


def test_ctypes_model_cfg_mirrors_the_header_struct_field_by_field():
    """lz_model_cfg crosses the boundary by value of its layout: the ctypes mirror (lightzero_amd/_lib.py) must list the header's fields in
    the header's order with the header's types -- a field added on one side only shifts every field behind it silently."""
    from lightzero_amd import _lib as L
    src = open(os.path.join(ROOT, "include", "lz_mi355.h")).read()
    body = re.search(r"typedef struct lz_model_cfg \{(.*?)\} lz_model_cfg;", src, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        typ, names = decl.split(None, 1)
        assert typ in ("int", "float"), decl
        fields += [(n.strip(), typ) for n in names.split(",")]
    want = {"int": ctypes.c_int, "float": ctypes.c_float}
    mirror = [(n, t) for n, t in L.ModelCfg._fields_]
    assert [n for n, _ in mirror] == [n for n, _ in fields]
    assert all(t is want[ht] for (_, t), (_, ht) in zip(mirror, fields))
    assert ctypes.sizeof(L.ModelCfg) == 4 * len(fields)
