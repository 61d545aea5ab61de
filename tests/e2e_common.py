"""End-to-end comparison of a fused device search with the oracle pipeline (reference-style driver + torch fp32 model + CPU ctree), shared
by the per-family tests: the fraction of roots with identical visit distributions is RECORDED (profiles/rNN_parity.json, entries
"e2e/<family>/...") and every differing root is attributed to its first differing selection (tests/e2e_attrib.py) -- a near-tie of two
pUCT scores or one quantum of the post-h^-1 scalars; an "unexplained" root fails the test.  Round 5 did this for configs[1] only
(tests/test_e2e_cfg1_gpu.py); VERDICT r5 weak #2 asked for the same record and a gate at the evidence for MuZero Atari, Go 9x9 and the
64x64 shapes instead of their 0.8 / 0.85 / 0.9 thresholds.
mcts_ctree.py:267-368, 745-876 (the drivers), cnode.cpp:651-695 / 756-814 (the selection)."""
import json

import numpy as np

import e2e_attrib
import parity_record


def device_records(roots, lib, L, B, A, S):
    """per-simulation records of a traced device search in the oracle drivers' record format"""
    trace = np.zeros((S, B, 4), np.int32)
    L.check(lib.lz_roots_read_trace(roots._h, S, trace.reshape(-1)))
    rec = []
    for s in range(S):
        vp = np.zeros(B, np.float32); val = np.zeros(B, np.float32); pol = np.zeros((B, A), np.float32)
        L.check(lib.lz_roots_read_sim_outputs(roots._h, s + 1, vp, val, pol.reshape(-1)))
        rec.append(dict(ix=trace[s, :, 0].copy(), action=trace[s, :, 1].copy(), search_len=trace[s, :, 2].copy(), value_prefix=vp, value=val, policy_logits=pol))
    return rec


def attribute_and_gate(name, kind, otree, cfg, A, legal, noises, to_play, o_logits, d_logits, rec_o, rec_d, o_dist, d_dist, o_val, d_val, gate):
    """records entry ``name``; asserts identical_fraction >= gate and that no differing root is unexplained.  Returns the summary."""
    B = len(o_dist)
    same = np.array([a == b for a, b in zip(o_dist, d_dist)])
    entries, benign = [], 0
    for b in range(B):
        e = e2e_attrib.attribute(otree, cfg, A, legal[b], None if noises is None else noises[b], o_logits[b], d_logits[b], rec_o, rec_d, b,
                                 kind=kind, to_play=to_play[b])
        if e is None:
            assert same[b], "root %d: every selection coincides and the visit distributions differ" % b
            continue
        if same[b]:
            benign += 1
        else:
            entries.append(e)
    summ = e2e_attrib.summarize(entries, B, int(same.sum()))
    summ["roots_with_a_differing_selection_but_identical_visit_counts"] = benign
    if same.any():
        summ["root_value_max_abs_diff_on_identical_roots"] = float(np.abs(np.asarray(o_val, np.float64) - np.asarray(d_val, np.float64))[same].max())
    summ["differing_roots"] = entries
    depth = np.stack([np.asarray(r["search_len"]) for r in rec_d])
    summ["search_depth_mean"], summ["search_depth_max"] = float(depth.mean()), int(depth.max())
    summ["gate"] = gate
    print(json.dumps({k: v for k, v in summ.items() if k != "differing_roots"}))
    for e in entries:
        print("  root %(root)d: first differing simulation %(first_sim)d, level %(level)d, gaps %(gap_oracle).3g / %(gap_device).3g, "
              "best two (oracle) %(best_two_oracle).3g, scalar delta %(scalar_delta).3g, range %(minmax_range).3g -> %(class)s" % e)
    parity_record.record(name, {}, extra=summ)
    assert same.mean() >= gate, "%s: only %d / %d roots have identical visit distributions (gate %.3f)" % (name, int(same.sum()), B, gate)
    bad = [e for e in entries if e["class"] == "unexplained"]
    assert not bad, "%s: differing roots not explained by a near-tie or one quantum of the post-transform scalars: %r" % (name, bad)
    return summ
