"""Attribution of end-to-end differences between two EfficientZero search pipelines that ran the same roots (the device engine and
the oracle pipeline = reference-style driver + torch fp32 model + reference ctree): for a root whose visit distributions differ,
find the FIRST simulation whose selection differed and say how close the competing pUCT scores were on both sides.

Both sides are given as per-simulation records (``ix`` = latent index of the leaf's parent, ``action``, ``search_len``, and the
network outputs ``value_prefix`` / ``value`` / ``policy_logits`` the tree was fed).  Up to the first differing selection the two
trees have the same shape, so each side's tree is rebuilt for that one root by replaying its own records through the CPU oracle
tree (oracle/ctree.py; the exact-replay gate shows the device tree is bit-equal to it on identical inputs) and the selection that
comes next is probed level by level without touching the tree (``Roots.probe`` -> ``otree_probe``: cucb_score, cnode.cpp:756-814,
of every legal action at every node of the walk).

Per differing root:
  first_sim        first simulation whose (ix, action) differ
  level            depth of the node where the two walks part (0 = root)
  gap_oracle       score(oracle's action) - score(device's action) in the ORACLE-side tree   (>= 0)
  gap_device       score(device's action) - score(oracle's action) in the DEVICE-side tree   (>= 0)
  best_two_oracle  best - second best score at that node, oracle side   ("best-two UCB scores within 1e-5")
  scalar_delta     max |value or value_prefix difference| fed to the two trees for this root before first_sim (post h^-1: the reference
                   formula's quantum is ~1.3e-4)
  minmax_range     max(value_delta_max, max - min) of the root's min-max statistics, oracle side: scores carry Q / range
  logit_delta      max |policy logit difference| fed to the two trees for this root before first_sim (priors)
  class            "near_tie" (both gaps <= 1e-5) | "quantum" (the score shift gap_oracle + gap_device is within 4 x scalar_delta / range + 16 x logit_delta:
                   a one-step difference of the post-transform scalars, amplified by the min-max normalisation, explains the flip) |
                   "unexplained" (flagged: would be a bug)
"""
import numpy as np


def first_difference(rec_a, rec_b, b):
    for s, (ra, rb) in enumerate(zip(rec_a, rec_b)):
        if int(ra["ix"][b]) != int(rb["ix"][b]) or int(ra["action"][b]) != int(rb["action"][b]):
            return s
    return None


def _replay_one(tree, cfg, A, S, legal, noise, root_logits, rec, b, upto, kind="ez", to_play=-1):
    """kind "ez": EfficientZero tree (value prefixes, reset flags); "mz": MuZero tree (rewards under the records' value_prefix key;
    to_play 1 | 2 in two-player mode)"""
    roots = tree.Roots(1, [list(legal)], action_space_size=A, max_simulations=S)
    roots.set_tiebreak(0)
    if noise is not None:
        roots.prepare(cfg["root_noise_weight"], [list(noise)], [0.0], [list(map(float, root_logits))], [int(to_play)])
    else:
        roots.prepare_no_noise([0.0], [list(map(float, root_logits))], [int(to_play)])
    mm = tree.MinMaxStatsList(1)
    mm.set_delta(cfg["value_delta_max"])
    for s in range(upto):
        res = tree.ResultsWrapper(1)
        ix, iy, la, vtp = tree.batch_traverse(roots, cfg["pb_c_base"], cfg["pb_c_init"], cfg["discount_factor"], mm, res, [int(to_play)])
        r = rec[s]
        assert ix[0] == int(r["ix"][b]) and la[0] == int(r["action"][b]), \
            "replaying root %d: simulation %d selected (%d, %d), the record says (%d, %d)" % (b, s, ix[0], la[0], r["ix"][b], r["action"][b])
        sl = res.get_search_len()[0]
        assert sl == int(r["search_len"][b])
        if kind == "ez":
            tree.batch_backpropagate(s + 1, cfg["discount_factor"], [float(r["value_prefix"][b])], [float(r["value"][b])],
                                     [list(map(float, r["policy_logits"][b]))], mm, res, [int(sl % cfg["lstm_horizon_len"] == 0)], vtp)
        else:
            tree.batch_backpropagate(s + 1, cfg["discount_factor"], [float(r["value_prefix"][b])], [float(r["value"][b])],
                                     [list(map(float, r["policy_logits"][b]))], mm, res, vtp)
    return roots


def attribute(tree, cfg, A, legal, noise, logits_oracle, logits_device, rec_oracle, rec_device, b, kind="ez", to_play=-1):
    """see the module docstring; returns None when every selection of root b coincides"""
    S = len(rec_oracle)
    s0 = first_difference(rec_oracle, rec_device, b)
    if s0 is None:
        return None
    ro = _replay_one(tree, cfg, A, S, legal, noise, logits_oracle, rec_oracle, b, s0, kind, to_play)
    rd = _replay_one(tree, cfg, A, S, legal, noise, logits_device, rec_device, b, s0, kind, to_play)
    args = (0, cfg["pb_c_base"], cfg["pb_c_init"], cfg["discount_factor"], 1 if int(to_play) == -1 else 2)
    po, pd = ro.probe(*args), rd.probe(*args)
    # each probe must end in the selection its own side recorded for simulation s0
    assert po[-1][0] == int(rec_oracle[s0]["ix"][b]) and po[-1][2] == int(rec_oracle[s0]["action"][b]), (po[-1], rec_oracle[s0]["ix"][b])
    assert pd[-1][0] == int(rec_device[s0]["ix"][b]) and pd[-1][2] == int(rec_device[s0]["action"][b]), (pd[-1], rec_device[s0]["ix"][b])
    lv = 0
    while po[lv][2] == pd[lv][2]:
        lv += 1
    assert po[lv][0] == pd[lv][0], "the walks parted before level %d" % lv
    so, sd = po[lv][1], pd[lv][1]
    ao, ad = po[lv][2], pd[lv][2]
    gap_o, gap_d = so[ao] - so[ad], sd[ad] - sd[ao]
    top = sorted(so.values(), reverse=True)
    delta = 0.0
    ldelta = float(np.max(np.abs(np.asarray(logits_oracle, np.float64) - np.asarray(logits_device, np.float64))))
    for s in range(s0):
        delta = max(delta, abs(float(rec_oracle[s]["value"][b]) - float(rec_device[s]["value"][b])),
                    abs(float(rec_oracle[s]["value_prefix"][b]) - float(rec_device[s]["value_prefix"][b])))
        ldelta = max(ldelta, float(np.max(np.abs(np.asarray(rec_oracle[s]["policy_logits"][b], np.float64) - np.asarray(rec_device[s]["policy_logits"][b], np.float64)))))
    mn, mx = ro.get_minmax()[0]
    rng = max(cfg["value_delta_max"], float(mx) - float(mn)) if mx > mn else 1.0
    shift = gap_o + gap_d
    if max(gap_o, gap_d) <= 1e-5:
        cls = "near_tie"
    elif shift <= 4.0 * delta / rng + 16.0 * ldelta + 1e-5:   # (priors move with the logits; pb_c sqrt(N) / (n + 1) stays below ~10 here)
        cls = "quantum"
    else:
        cls = "unexplained"
    return dict(root=int(b), first_sim=int(s0), level=int(lv), node_latent=int(po[lv][0]), action_oracle=int(ao), action_device=int(ad),
                gap_oracle=float(gap_o), gap_device=float(gap_d), best_two_oracle=float(top[0] - top[1]) if len(top) > 1 else None,
                scalar_delta=float(delta), logit_delta=float(ldelta), minmax_range=float(rng), score_shift=float(shift), **{"class": cls})


def summarize(entries, n_roots, n_same):
    cls = {}
    for e in entries:
        cls[e["class"]] = cls.get(e["class"], 0) + 1
    return dict(roots=int(n_roots), identical_visit_distributions=int(n_same), identical_fraction=n_same / float(n_roots),
                differing=len(entries), classes=cls,
                worst_gap=max([max(e["gap_oracle"], e["gap_device"]) for e in entries] or [0.0]),
                note="every differing root: the first simulation whose selection differed, the pUCT score gaps of the two competing "
                     "actions on both sides, and whether a near-tie (<= 1e-5) or a one-quantum difference of the post-h^-1 scalars "
                     "(scaled by the min-max range) explains the flip; 'unexplained' would be a bug")
