"""Action spaces beyond 256 (csrc/lz_tree_wide.hip; Chinese chess: 2086 moves,
zoo/board_games/chinese_chess/config/chinese_chess_muzero_bot_mode_config.py:33).

The wide cases of tests/tree_driver.py::CASES are held bit-exact to the C oracle and to the goldens of the compiled reference by
tests/test_tree_gpu.py like every other case.  Here: (i) the wide kernels against the register kernels on action spaces BOTH serve
(LZ_TREE_WIDE=1 sends every MuZero / EfficientZero tree through lz_tree_wide.hip), in both tie-break modes -- the stochastic draw is
keyed by (seed, epoch, traverse counter, root, depth) and must pick the same member of the same tie list; (ii) what stays refused."""
import os

import numpy as np
import pytest

import tree_driver as td

pytestmark = pytest.mark.gpu


def _mod(variant):
    if variant == "ez":
        from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree as m
    else:
        from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree as m
    return m


def _run(c, tiebreak, wide):
    mod = _mod(c["variant"])
    orig = mod.Roots

    def mk(n, legal, **kw):
        r = orig(n, legal, action_space_size=c["A"], max_simulations=c["S"])
        r.set_tiebreak(tiebreak, seed=1234)
        return r
    ns = type("M", (), dict(Roots=staticmethod(mk), MinMaxStatsList=mod.MinMaxStatsList, ResultsWrapper=mod.ResultsWrapper,
                            batch_traverse=staticmethod(mod.batch_traverse), batch_backpropagate=staticmethod(mod.batch_backpropagate)))
    if wide:
        os.environ["LZ_TREE_WIDE"] = "1"
    else:
        os.environ.pop("LZ_TREE_WIDE", None)
    try:
        return td.run_tree(ns, c)
    finally:
        os.environ.pop("LZ_TREE_WIDE", None)


NARROW = ["ez_probe_b4", "ez_fixture16_2p", "ez_zero_ties", "ez_big_a20", "mz_go82_2p", "mz_zero_ties", "ez_wide_a150", "mz_wide_a200_2p",
          "mz_runaway_logits", "ez_runaway_logits", "ez_deep_chain", "mz_deep_chain", "mz_fixture16_2p"]


@pytest.mark.parametrize("tiebreak", [0, 1])
@pytest.mark.parametrize("name", NARROW)
def test_wide_kernels_are_bit_identical_to_the_register_kernels(name, tiebreak):
    c = td.make_inputs(td.CASES[name])
    a = _run(c, tiebreak, wide=False)
    b = _run(c, tiebreak, wide=True)
    td.assert_same(a, b, name)
    assert np.array_equal(a["minmax"].view(np.uint32), b["minmax"].view(np.uint32)), "min/max stats differ"


@pytest.mark.parametrize("name", ["mz_wide_a300_zero_ties", "ez_go19_a362"])
def test_stochastic_tie_break_on_wide_nodes_draws_from_the_tie_list(name):
    """All-zero networks: every unvisited child of a node ties.  The deterministic mode takes the first of them (held to the
    reference by test_tree_gpu.py); the stochastic mode must spread over the list, stay inside the legal list, and be reproducible."""
    c = td.make_inputs(dict(td.CASES[name], zero=True, noise_w=None))
    r1 = _run(c, 1, wide=False)
    r2 = _run(c, 1, wide=False)
    td.assert_same(r1, r2, name)
    first = r1["records"][:, :, 2]        # action chosen at the end of every simulation's path
    for b in range(c["B"]):
        legal = set(c["legal_list"][b])
        roots_actions = [int(a) for a, d in zip(first[:, b], r1["records"][:, b, 3]) if d == 1]
        assert set(roots_actions) <= legal
        assert len(set(roots_actions)) == len(roots_actions), "an unvisited root child was selected twice while others were unvisited"
        assert roots_actions != sorted(roots_actions), "the draw never left list order"
        assert max(roots_actions) >= 64, "no selection beyond the first 64-lane chunk"


def test_what_stays_refused_beyond_256_actions():
    import ctypes
    from lightzero_amd import _lib as L
    lib = L.lib()
    eng = L.default_engine()
    h = L.P()
    rc = lib.lz_roots_create(eng, 3, 1, 1100, 8, L.i32(list(range(1100))), L.i32([1100]), ctypes.byref(h))   # the Gumbel tree: 16 register chunks
    assert rc < 0 and "Gumbel" in lib.lz_last_error().decode()
    rc = lib.lz_roots_create(eng, 1, 1, 70000, 8, L.i32(list(range(70000))), L.i32([70000]), ctypes.byref(h))
    assert rc < 0 and "65535" in lib.lz_last_error().decode()


def _run_reuse(c, wide):
    mod = _mod(c["variant"])
    orig = mod.Roots

    def mk(n, legal, **kw):
        r = orig(n, legal, action_space_size=c["A"], max_simulations=c["S"])
        r.set_tiebreak(0)
        return r
    ns = type("M", (), dict(Roots=staticmethod(mk), MinMaxStatsList=mod.MinMaxStatsList, ResultsWrapper=mod.ResultsWrapper,
                            batch_traverse_with_reuse=staticmethod(mod.batch_traverse_with_reuse),
                            batch_backpropagate_with_reuse=staticmethod(mod.batch_backpropagate_with_reuse)))
    if wide:
        os.environ["LZ_TREE_WIDE"] = "1"
    try:
        return td.run_tree_reuse(ns, c)
    finally:
        os.environ.pop("LZ_TREE_WIDE", None)


@pytest.mark.parametrize("name", sorted(td.REUSE_CASES))
def test_rezero_reuse_surface_on_the_wide_kernels(name):
    """batch_traverse_with_reuse / batch_backpropagate_with_reuse (cnode.cpp:603-649, 965-1072): the cases both kernel sets serve are
    bit-identical between them; every case -- the two beyond 256 actions included -- is bit-exact to the C oracle and to the golden of the
    reference's compiled ctree"""
    from oracle import ctree as octree
    c = td.make_reuse_inputs(td.REUSE_CASES[name])
    wide = _run_reuse(c, wide=True)
    if c["A"] <= 256:
        narrow = _run_reuse(c, wide=False)
        td.assert_same(narrow, wide, name)
        assert narrow["inferences"] == wide["inferences"]
    omod = octree.ez_tree if c["variant"] == "ez" else octree.mz_tree
    ora = td.run_tree_reuse(omod, c, roots_kwargs=dict(action_space_size=c["A"], max_simulations=c["S"]))
    td.assert_same(ora, wide, name)
    assert ora["inferences"] == wide["inferences"]
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tree_%s.npz" % name))
    assert np.array_equal(wide["records"], g["records"])
    assert np.array_equal(wide["values"].view(np.uint32), g["values"].view(np.uint32))


def _timed_tree_loop(mod, c, roots, as_lists):
    """the MuZero driver's per-simulation calls (mcts_ctree.py:300-368) with recorded network outputs; only batch_traverse /
    batch_backpropagate are inside the clock.  as_lists: the reference's Cython surface takes Python lists (what its driver builds with
    .tolist(), not timed here); the device side takes the numpy rows lightzero_amd's foreign-model loop hands over"""
    import time
    B, S = c["B"], c["S"]
    mm = mod.MinMaxStatsList(B)
    mm.set_delta(c["delta"])
    rec = np.zeros((S, B, 5), np.int32)
    conv = (lambda a: a.tolist()) if as_lists else (lambda a: a)
    sims = [(conv(x["vp"]), conv(x["v"]), conv(x["logits"])) for x in c["sims"]]
    t = 0.0
    for s in range(S):
        res = mod.ResultsWrapper(B)
        tp = list(c["to_play_list"])
        t0 = time.perf_counter()
        if as_lists:
            ix, iy, la, vtp = mod.batch_traverse(roots, c["pb_c_base"], c["pb_c_init"], c["discount"], mm, res, tp, deterministic=True)
        else:
            ix, iy, la, vtp = mod.batch_traverse(roots, c["pb_c_base"], c["pb_c_init"], c["discount"], mm, res, tp)
        t += time.perf_counter() - t0
        rec[s, :, 0], rec[s, :, 1], rec[s, :, 2], rec[s, :, 3], rec[s, :, 4] = ix, iy, la, res.get_search_len(), vtp
        vp, v, lg = sims[s]
        t0 = time.perf_counter()
        mod.batch_backpropagate(s + 1, c["discount"], vp, v, lg, mm, res, vtp)
        t += time.perf_counter() - t0
    return dict(records=rec, distributions=roots.get_distributions(), values=np.asarray(roots.get_values(), np.float32)), t


@pytest.mark.parametrize("B", [8, 256])
def test_chinese_chess_sized_search_matches_the_compiled_reference_and_is_timed(B):
    """2086 actions, ~40 legal at a root, 50 simulations (the preset's collector runs 8 environments; 256 = BASELINE's batch): the device
    tree behind the fine-grained API against the reference's compiled ctree (oracle/_ref/det when the box has it, else the C oracle)
    on the same recorded network outputs -- identical records / visit counts / bit-equal values -- with both sides' time inside
    batch_traverse + batch_backpropagate per simulation written to gpurun_out/tree_wide_timing.json (committed copy:
    profiles/r06_tree_wide_timing.json)."""
    import json
    from oracle import build_ref, ctree as octree
    from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree
    c = td.make_inputs(dict(td.CASES["mz_xiangqi_a2086_2p"], B=B, seed=40 + B))
    mods = build_ref.load("det")
    kind = "reference (oracle/_ref/det)" if mods else "C oracle"
    if mods:
        rr = mods[1].Roots(B, c["legal_list"])
    else:
        rr = octree.mz_tree.Roots(B, c["legal_list"], action_space_size=c["A"], max_simulations=c["S"])
    rr.prepare(c["noise_w"], c["noises"], c["root_vp"].tolist(), c["root_logits"].tolist(), list(c["to_play_list"]))
    if mods:
        ref, t_ref = _timed_tree_loop(mods[1], c, rr, True)
    else:   # (the C oracle's Python surface has no `deterministic` keyword: its ties are deterministic by construction)
        class _O(object):
            MinMaxStatsList, ResultsWrapper, batch_backpropagate = octree.mz_tree.MinMaxStatsList, octree.mz_tree.ResultsWrapper, staticmethod(octree.mz_tree.batch_backpropagate)
            batch_traverse = staticmethod(lambda *a, **k: octree.mz_tree.batch_traverse(*a))
        ref, t_ref = _timed_tree_loop(_O, c, rr, True)
    t_dev = None
    for _ in range(2):   # the first pass warms allocations and the handle cache
        dr = mz_tree.Roots(B, c["legal_list"], action_space_size=c["A"], max_simulations=c["S"])
        dr.set_tiebreak(0)
        dr.prepare(c["noise_w"], c["noises"], c["root_vp"].tolist(), c["root_logits"].tolist(), list(c["to_play_list"]))
        dev, t_dev = _timed_tree_loop(mz_tree, c, dr, False)
    td.assert_same(ref, dev, "xiangqi B=%d" % B)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "tree_wide_timing.json")
    try:
        data = json.load(open(path))
    except Exception:
        data = {}
    data["what"] = ("MuZero tree, A = 2086, 2 % legal at the root, 50 simulations, fine-grained API: wall ms inside batch_traverse + batch_backpropagate per "
                    "simulation for all roots (device: numpy rows in, pinned staging and PCIe included; host tree: Python lists in, as its Cython surface requires)")
    data.setdefault("rows", {})["B=%d" % B] = {"device_ms_per_simulation": 1e3 * t_dev / c["S"], "host_tree_ms_per_simulation": 1e3 * t_ref / c["S"],
                                               "host_tree": kind, "identical": True}
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)


def test_baseline_sized_searches_replay_exactly_through_the_wide_kernels():
    """LZ_TREE_WIDE=1 LZ_NO_TREE_FUSE=1: every tree step of the fused search is a separate launch of lz_tree_wide.hip's expand + backup +
    selection kernel.  The exact-replay suite at BASELINE sizes (configs[1] 256 x 50, configs[2] 1024 x 400, configs[3] Go 64 x 200 two
    players, ragged batches) must still reproduce the reference's compiled ctree -- in a process of its own, like the other switches."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LZ_TREE_WIDE="1", LZ_NO_TREE_FUSE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_exact_replay_gpu.py"), "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "configs1_full or configs2 or configs3 or gomoku or odd_batches or muzero_atari"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "LZ_TREE_WIDE=1:\n%s\n%s" % (r.stdout[-3000:], r.stderr[-2000:])
    assert " passed" in r.stdout


_OFF = int(os.environ.get("LZ_FUZZ_SEED_OFFSET", "0"))


def _wide_case(seed):
    r = np.random.default_rng(9500 + seed)
    two = bool(r.integers(0, 2))
    return dict(variant=["ez", "mz"][int(r.integers(0, 2))], B=int(r.integers(1, 7)), A=int(r.integers(257, 2600)), S=int(r.integers(1, 49)),
                seed=300 + seed, legal=[None, "random"][int(r.integers(0, 2))], legal_p=float(r.choice([0.01, 0.1, 0.6])),
                to_play="random12" if two else None, discount=float(r.choice([0.997, 1.0, 0.9])), pb_c_base=int(r.choice([19652, 1, 100])),
                pb_c_init=float(r.choice([1.25, 0.5, 2.0])), delta=float(r.choice([0.01, 0.0, 0.1])), noise_w=[0.25, None, 0.5][int(r.integers(0, 3))],
                horizon=int(r.choice([5, 1, 3])), scale=float(r.choice([1.0, 5.0, 0.1])), zero=bool(r.random() < 0.1))


@pytest.mark.parametrize("seed", range(_OFF, _OFF + 16))
def test_random_wide_configuration_matches_the_oracle(seed):
    """the randomised sweep of tests/test_tree_fuzz_gpu.py on action spaces of 257 .. 2599: both tree variants, full and ragged legal lists (1 % .. 60 %
    legal), one and two players, discounts, pb_c constants, min-max deltas, noise weights, LSTM horizons, logit scales -- bit-exact against the C oracle
    (records, visit counts, root values, min-max statistics)"""
    from oracle import ctree as octree
    case = _wide_case(seed)
    c = td.make_inputs(case)
    dev = _run(c, 0, wide=False)
    omod = octree.ez_tree if c["variant"] == "ez" else octree.mz_tree
    ora = td.run_tree(omod, c, roots_kwargs=dict(action_space_size=c["A"], max_simulations=c["S"]))
    td.assert_same(ora, dev, repr(case))
    assert np.array_equal(ora["minmax"].view(np.uint32), dev["minmax"].view(np.uint32)), "min/max stats differ: %r" % (case,)
