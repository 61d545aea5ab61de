"""Shared driver for the tree-only parity tests: steps any module exposing the reference's Cython
surface (``Roots``, ``MinMaxStatsList``, ``ResultsWrapper``, ``batch_traverse``, ``batch_backpropagate``;
lzero/mcts/ctree/ctree_efficientzero/ez_tree.pyx, ctree_muzero/mz_tree.pyx) through S simulations
with *recorded* (seeded random) network outputs, the way EfficientZeroMCTSCtree.search does
(lzero/mcts/tree_search/mcts_ctree.py:782-876), and records everything that is observable.
"""
import numpy as np

# the 16-env x 9-action legal-action fixture of lzero/mcts/tests/test_mcts_ctree.py:122-152
FIXTURE_ACTION_MASK = [
    [0, 0, 0, 1, 0, 1, 1, 0, 0], [1, 0, 0, 1, 0, 0, 1, 0, 0], [1, 1, 0, 0, 1, 0, 1, 0, 1], [1, 0, 0, 1, 1, 1, 0, 0, 0],
    [0, 0, 1, 0, 0, 1, 0, 0, 1], [0, 1, 1, 0, 1, 0, 0, 0, 0], [1, 0, 1, 1, 1, 0, 0, 1, 1], [1, 1, 1, 1, 1, 0, 0, 0, 1],
    [0, 0, 0, 1, 0, 1, 1, 0, 0], [0, 1, 1, 0, 1, 1, 1, 1, 0], [1, 1, 1, 0, 0, 0, 1, 1, 1], [1, 1, 0, 1, 0, 1, 1, 0, 0],
    [0, 0, 1, 0, 0, 1, 0, 0, 0], [1, 0, 1, 1, 0, 0, 1, 1, 0], [0, 1, 0, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 1, 1, 0, 0, 1],
]
FIXTURE_TO_PLAY = [2, 1, 2, 1, 1, 2, 2, 1, 1, 1, 2, 1, 2, 1, 1, 1]


def fixture_legal_actions():
    return [np.nonzero(np.array(m))[0].tolist() for m in FIXTURE_ACTION_MASK]


CASES = {
    # name: dict(variant, B, A, S, seed, legal, to_play, pb_c_base, pb_c_init, discount, delta, noise_w, horizon, scale)
    "ez_probe_b4": dict(variant="ez", B=4, A=6, S=50, seed=0),
    "ez_cfg2_b256": dict(variant="ez", B=256, A=6, S=50, seed=1),
    "ez_fixture16": dict(variant="ez", B=16, A=9, S=8, seed=2, legal="fixture", to_play=[-1] * 16, pb_c_base=1,
                         pb_c_init=1.0, discount=0.9, noise_w=0.2),
    "ez_fixture16_2p": dict(variant="ez", B=16, A=9, S=40, seed=3, legal="fixture", to_play=FIXTURE_TO_PLAY,
                            discount=1.0),
    "ez_zero_ties": dict(variant="ez", B=8, A=6, S=30, seed=4, zero=True, noise_w=None),
    "ez_big_a20": dict(variant="ez", B=32, A=20, S=100, seed=5),
    "mz_cartpole": dict(variant="mz", B=8, A=2, S=25, seed=6),
    "mz_breakout": dict(variant="mz", B=64, A=4, S=400, seed=7),
    "mz_go82_2p": dict(variant="mz", B=16, A=82, S=200, seed=8, legal="random", to_play="random12", discount=1.0),
    "mz_fixture16_2p": dict(variant="mz", B=16, A=9, S=40, seed=9, legal="fixture", to_play=FIXTURE_TO_PLAY),
    "mz_zero_ties": dict(variant="mz", B=4, A=3, S=12, seed=10, zero=True, noise_w=None),
    # values that grow with the simulation index: the newest leaf always looks best, so the search keeps extending ONE path
    # (search depth well beyond 64: path records longer than a wavefront, deep backups)
    # action spaces beyond 128: four 64-lane chunks per node in the device kernels
    "ez_wide_a150": dict(variant="ez", B=6, A=150, S=60, seed=13, legal="random"),
    "mz_wide_a200_2p": dict(variant="mz", B=5, A=200, S=80, seed=14, legal="random", to_play="random12", discount=1.0),
    # action spaces beyond 256 (lz_tree_wide.hip: a node's children walked in 64-lane chunks by a loop).  Chinese chess has 2086 moves
    # of which a few dozen are legal at a root (zoo/board_games/chinese_chess/config/chinese_chess_muzero_bot_mode_config.py:33) --
    # below the root every action is "legal" (cnode.cpp:88-151); Go 19x19 has 362
    "mz_xiangqi_a2086_2p": dict(variant="mz", B=8, A=2086, S=50, seed=30, legal="random", legal_p=0.02, to_play="random12", discount=1.0),
    "ez_go19_a362": dict(variant="ez", B=6, A=362, S=60, seed=31, legal="random"),
    "mz_wide_a300_zero_ties": dict(variant="mz", B=4, A=300, S=40, seed=32, zero=True, noise_w=None),
    "ez_wide_a1000_runaway": dict(variant="ez", B=6, A=1000, S=40, seed=33, runaway=True),
    "ez_wide_a257_sharp": dict(variant="ez", B=5, A=257, S=70, seed=34, scale=4.0),
    "mz_wide_a65535_max": dict(variant="mz", B=2, A=65535, S=6, seed=35, legal="random", legal_p=0.001),   # the largest action space the node links hold
    # a diverged network (random weights unrolled 40+ steps deep produce logits of 1e14 and h^-1 values of 1e4): nodes whose
    # logits all lie below FLOAT_MIN = -1e6 get priors 0 / 0 = NaN (cnode.cpp:123-137), every score at such a node is NaN, no child
    # enters the tie list and cselect_child returns its default action 0 (cnode.cpp:687-693); found by the exact replay gate at
    # BASELINE configs[2] size (round 2)
    "mz_runaway_logits": dict(variant="mz", B=24, A=4, S=120, seed=15, runaway=True),
    "ez_runaway_logits": dict(variant="ez", B=24, A=6, S=120, seed=16, runaway=True),
    "ez_deep_chain": dict(variant="ez", B=4, A=2, S=150, seed=11, deep=True),
    "mz_deep_chain": dict(variant="mz", B=3, A=3, S=200, seed=12, deep=True),
}


def make_inputs(case):
    """Seeded synthetic 'network outputs' for every simulation of a case (all float32)."""
    c = dict(pb_c_base=19652, pb_c_init=1.25, discount=0.997, delta=0.01, noise_w=0.25, horizon=5, legal=None,
             to_play=None, zero=False, scale=1.0, deep=False, runaway=False, legal_p=0.6)
    c.update(case)
    rng = np.random.default_rng(c["seed"])
    B, A, S = c["B"], c["A"], c["S"]
    if c["legal"] == "fixture":
        legal = fixture_legal_actions()
    elif c["legal"] == "random":
        legal = []
        for _ in range(B):
            m = rng.random(A) < c["legal_p"]
            m[A - 1] = True  # "pass" always legal
            legal.append(np.nonzero(m)[0].tolist())
    else:
        legal = [list(range(A)) for _ in range(B)]
    if c["to_play"] is None:
        to_play = [-1] * B
    elif c["to_play"] == "random12":
        to_play = rng.integers(1, 3, size=B).tolist()
    else:
        to_play = list(c["to_play"])
    z = 0.0 if c["zero"] else 1.0
    root_logits = (z * c["scale"] * rng.standard_normal((B, A))).astype(np.float32)
    if c["deep"]:
        root_logits[:, 0] += 12.0
    root_vp = np.zeros(B, np.float32)  # initial_inference: value_prefix = [0.]*B (efficientzero_model.py:238)
    if c["variant"] == "mz":
        root_vp = (z * 0.1 * rng.standard_normal(B)).astype(np.float32)  # MuZero roots carry a reward
    noises = None
    if c["noise_w"] is not None:
        noises = [rng.dirichlet([0.3] * len(l)).astype(np.float32).tolist() for l in legal]
    sims = []
    for si in range(S):
        sims.append(dict(
            vp=(z * 0.5 * rng.standard_normal(B)).astype(np.float32),
            v=(z * rng.standard_normal(B)).astype(np.float32),
            logits=(z * c["scale"] * rng.standard_normal((B, A))).astype(np.float32),
        ))
        if c["runaway"]:
            bad = rng.random(B) < 0.35                     # these leaves: every logit far below -1e6
            mag = (10.0 ** rng.uniform(7, 15, size=(B, 1))).astype(np.float32)
            lg = sims[-1]["logits"]
            lg[bad] = (-mag * (1.0 + rng.random((B, A)).astype(np.float32)))[bad]
            hot = rng.random(B) < 0.2                      # and some with one overwhelming action
            lg[hot & ~bad, 0] = 1e13
            sims[-1]["vp"] = (sims[-1]["vp"] * 2e4).astype(np.float32)
            sims[-1]["v"] = (sims[-1]["v"] * 2e4).astype(np.float32)
        if c["deep"]:
            sims[-1]["vp"] = (0.01 * rng.random(B)).astype(np.float32)
            sims[-1]["v"] = (1.0 + 0.5 * si + 0.01 * rng.random(B)).astype(np.float32)
            lg = (0.01 * rng.standard_normal((B, A))).astype(np.float32)
            lg[:, 0] += 12.0   # one overwhelmingly likely action per node
            sims[-1]["logits"] = lg
    c.update(legal_list=legal, to_play_list=to_play, root_logits=root_logits, root_vp=root_vp, noises=noises, sims=sims)
    return c


def run_tree(mod, c, roots_kwargs=None, traverse_kwargs=None):
    """Returns dict(records=int32 [S,B,5] (ix,iy,last_action,search_len,virtual_to_play),
    distributions=list[list[int]], values=float32 [B])."""
    B, S = c["B"], c["S"]
    ez = c["variant"] == "ez"
    roots = mod.Roots(B, c["legal_list"], **(roots_kwargs or {}))
    if c["noises"] is not None:
        roots.prepare(c["noise_w"], c["noises"], c["root_vp"].tolist(), c["root_logits"].tolist(), list(c["to_play_list"]))
    else:
        roots.prepare_no_noise(c["root_vp"].tolist(), c["root_logits"].tolist(), list(c["to_play_list"]))
    mm = mod.MinMaxStatsList(B)
    mm.set_delta(c["delta"])
    rec = np.zeros((S, B, 5), np.int32)
    for s in range(S):
        res = mod.ResultsWrapper(B)
        ix, iy, la, vtp = mod.batch_traverse(roots, c["pb_c_base"], c["pb_c_init"], c["discount"], mm, res,
                                             list(c["to_play_list"]), **(traverse_kwargs or {}))
        sl = res.get_search_len()
        rec[s, :, 0], rec[s, :, 1], rec[s, :, 2], rec[s, :, 3], rec[s, :, 4] = ix, iy, la, sl, vtp
        sim = c["sims"][s]
        if ez:
            reset = [int(l % c["horizon"] == 0) for l in sl]
            mod.batch_backpropagate(s + 1, c["discount"], sim["vp"].tolist(), sim["v"].tolist(), sim["logits"].tolist(),
                                    mm, res, reset, vtp)
        else:
            mod.batch_backpropagate(s + 1, c["discount"], sim["vp"].tolist(), sim["v"].tolist(), sim["logits"].tolist(),
                                    mm, res, vtp)
    out = dict(records=rec, distributions=roots.get_distributions(),
               values=np.asarray(roots.get_values(), np.float32))
    if hasattr(roots, "get_minmax"):
        out["minmax"] = np.asarray(roots.get_minmax(), np.float32)
    return out


def assert_same(a, b, what=""):
    assert np.array_equal(a["records"], b["records"]), "%s: per-simulation (ix,iy,action,len,to_play) differ" % what
    assert a["distributions"] == b["distributions"], "%s: visit-count distributions differ" % what
    # bit-exact: same float32 op order on both sides
    assert np.array_equal(a["values"].view(np.uint32), b["values"].view(np.uint32)), "%s: root values differ" % what


# ------------------------------------------------------------------------------------------------
# ReZero: search_with_reuse (lzero/mcts/tree_search/mcts_ctree.py:878-1002 EZ, :370-470 MZ)
# ------------------------------------------------------------------------------------------------
REUSE_CASES = {
    "ez_reuse_b32": dict(variant="ez", B=32, A=6, S=40, seed=20),
    "ez_reuse_fixture16": dict(variant="ez", B=16, A=9, S=30, seed=21, legal="fixture", to_play=[-1] * 16),
    "ez_reuse_fixture16_2p": dict(variant="ez", B=16, A=9, S=30, seed=22, legal="fixture", to_play=FIXTURE_TO_PLAY, discount=1.0),
    "mz_reuse_b24": dict(variant="mz", B=24, A=4, S=60, seed=23),
    "mz_reuse_fixture16_2p": dict(variant="mz", B=16, A=9, S=30, seed=24, legal="fixture", to_play=FIXTURE_TO_PLAY),
    # action spaces beyond 256 (lz_tree_wide.hip)
    "mz_reuse_wide_a300_2p": dict(variant="mz", B=6, A=300, S=40, seed=25, legal="random", legal_p=0.01, to_play="random12", discount=1.0),
    "ez_reuse_wide_a1000": dict(variant="ez", B=5, A=1000, S=60, seed=27, legal="random", legal_p=0.004),
}


def make_reuse_inputs(case):
    c = make_inputs(case)
    rng = np.random.default_rng(1000 + c["seed"])
    # the trajectory's true action (a legal one) and the value found by the search of the next state
    c["true_action"] = [int(l[rng.integers(0, len(l))]) for l in c["legal_list"]]
    c["reuse_value"] = rng.standard_normal(c["B"]).astype(np.float32)
    return c


def run_tree_reuse(mod, c, roots_kwargs=None):
    """The reference driver's bookkeeping (compaction of the roots that need inference, no_inference / reuse lists with
    the -1 sentinel) around batch_traverse_with_reuse / batch_backpropagate_with_reuse, with recorded network outputs:
    sims[s] holds one row per ROOT; the rows of the roots that need inference are packed in root order.
    is_reset is computed per root from its own search length (the reference indexes a compacted list by root there,
    mcts_ctree.py:968-971 vs cnode.cpp:645, which reads out of bounds whenever a root skips inference)."""
    B, S = c["B"], c["S"]
    ez = c["variant"] == "ez"
    roots = mod.Roots(B, c["legal_list"], **(roots_kwargs or {}))
    if c["noises"] is not None:
        roots.prepare(c["noise_w"], c["noises"], c["root_vp"].tolist(), c["root_logits"].tolist(), list(c["to_play_list"]))
    else:
        roots.prepare_no_noise(c["root_vp"].tolist(), c["root_logits"].tolist(), list(c["to_play_list"]))
    mm = mod.MinMaxStatsList(B)
    mm.set_delta(c["delta"])
    rec = np.zeros((S, B, 5), np.int32)
    infer = 0
    for s in range(S):
        res = mod.ResultsWrapper(B)
        ix, iy, la, vtp = mod.batch_traverse_with_reuse(roots, c["pb_c_base"], c["pb_c_init"], c["discount"], mm, res,
                                                        list(c["to_play_list"]), list(c["true_action"]), c["reuse_value"].tolist())
        sl = res.get_search_len()
        rec[s, :, 0], rec[s, :, 1], rec[s, :, 2], rec[s, :, 3], rec[s, :, 4] = ix, iy, la, sl, vtp
        need, no_inference_lst, reuse_lst = [], [], []
        for count in range(B):
            if ix[count] != -1:
                need.append(count)
            else:
                no_inference_lst.append(iy[count])
            if ix[count] == 0 and la[count] == c["true_action"][count]:
                reuse_lst.append(count)
        infer += len(need)
        sim = c["sims"][s]
        no_inference_lst.append(-1)
        reuse_lst.append(-1)
        vp, v, lg = sim["vp"][need].tolist(), sim["v"][need].tolist(), sim["logits"][need].tolist()
        if ez:
            reset = [int(l % c["horizon"] == 0) for l in sl]
            mod.batch_backpropagate_with_reuse(s + 1, c["discount"], vp, v, lg, mm, res, reset, vtp, no_inference_lst, reuse_lst,
                                               c["reuse_value"].tolist())
        else:
            mod.batch_backpropagate_with_reuse(s + 1, c["discount"], vp, v, lg, mm, res, vtp, no_inference_lst, reuse_lst,
                                               c["reuse_value"].tolist())
    out = dict(records=rec, distributions=roots.get_distributions(), values=np.asarray(roots.get_values(), np.float32),
               inferences=infer)
    return out
