"""GameSegmentBatch.rollover -- the collector's segment hand-over (muzero_collector.py:308-410, 649-692: is_full ->
pad_over from the successor's first num_unroll_steps + td_steps entries -> pool with priorities and done; episode ends) -- against the
reference's OWN GameSegment class (pad_over, is_full, game_segment_to_array, imported from /root/reference) driven by the collector's
per-env logic restated here (the collector module itself imports DI-engine), with random episode ends, ragged legal actions, the
sampled / Gumbel extra fields and priorities.  Also ADVICE r2: a partial reset must leave the other envs' frames alone."""
import numpy as np
import pytest

import ref_loader
from lightzero_amd import shard
from lightzero_amd.mcts.buffer.game_segment import GameSegmentBatch

N, A, L, STACK, FRAME, UNROLL, TD = 5, 6, 6, 3, (1, 4, 4), 2, 3


def _cfg(**kw):
    from easydict import EasyDict
    return EasyDict(dict(num_unroll_steps=UNROLL, td_steps=TD, discount_factor=0.997, gray_scale=False, transform2string=False,
                         sampled_algo=kw.get("sampled", False), gumbel_algo=kw.get("gumbel", False), use_ture_chance_label_in_chance_encoder=False,
                         model=dict(frame_stack_num=STACK, action_space_size=A, image_channel=1, observation_shape=(STACK, 4, 4))))


@pytest.mark.parametrize("mode", ["plain", "ragged_priority", "gumbel", "sampled"])
def test_rollover_equals_the_collectors_logic_on_reference_segments(mode):
    ref = ref_loader.load()
    if ref is None:
        pytest.skip("/root/reference not present")
    GS = ref.game_segment.GameSegment
    ragged, use_pri = mode == "ragged_priority", mode == "ragged_priority"
    gumbel, sampled = mode == "gumbel", mode == "sampled"
    K, D = 4, 2
    rng = np.random.default_rng({"plain": 1, "ragged_priority": 2, "gumbel": 3, "sampled": 4}[mode])
    cfg = _cfg(sampled=sampled, gumbel=gumbel)
    AW = K if sampled else A            # width of the visit-count block: the K sampled actions for Sampled EfficientZero
    first = rng.random((N,) + FRAME).astype(np.float32)
    # ---- reference side: the collector's state (muzero_collector.py:470-510)
    window = [[first[e]] * STACK for e in range(N)]
    segs = [GS(None, game_segment_length=L, config=cfg) for _ in range(N)]
    for e in range(N):
        segs[e].reset(window[e])
    last, last_pri, pool = [None] * N, [None] * N, []
    pred_l, search_l = [[] for _ in range(N)], [[] for _ in range(N)]

    def priorities(e):   # _compute_priorities (:308-334)
        if not use_pri:
            return None
        return np.abs(np.asarray(pred_l[e], np.float32) - np.asarray(search_l[e], np.float32)) + np.float32(1e-6)

    def pad_and_save(e, done):   # pad_and_save_last_trajectory (:336-410)
        p = UNROLL + TD
        g = segs[e]
        last[e].valid_transition_count = min(len(last[e].action_segment), L)
        kw = dict(next_segment_improved_policy=g.improved_policy_probs[:p]) if gumbel else {}
        last[e].pad_over(g.obs_segment[STACK:STACK + p], g.reward_segment[:p - 1], g.action_segment[:p], g.root_value_segment[:p],
                         g.child_visit_segment[:p], **kw)
        last[e].game_segment_to_array()
        pool.append((last[e], last_pri[e], done))
        last[e], last_pri[e] = None, None
    # ---- engine side
    batch = GameSegmentBatch(N, AW, L, FRAME, frame_stack_num=STACK, num_unroll_steps=UNROLL, td_steps=TD, use_priority=use_pri,
                             sampled_actions_shape=(K, D) if sampled else None, improved_policy=gumbel, continuous_action_space=sampled)
    batch.reset(np.repeat(first[:, None], STACK, 1))
    for t in range(60):
        out, masks, tps, extra = {}, [], [], []
        for e in range(N):
            m = np.ones(AW, np.float32)
            if ragged:
                m = (rng.random(AW) < 0.6).astype(np.float32)
                m[rng.integers(0, AW)] = 1
            legal = np.nonzero(m)[0]
            visits = rng.integers(0, 20, size=len(legal)).tolist()
            ex = rng.random((K, D)).astype(np.float32) if sampled else (rng.random(A).astype(np.float32) if gumbel else None)
            pick = int(legal[rng.integers(0, len(legal))])
            # Sampled EfficientZero, continuous: the action IS the chosen sampled action, a [D] vector (sampled_efficientzero.py:905-907)
            out[e] = dict(action=ex[pick].copy() if sampled else pick, visit_count_distributions=visits,
                          visit_count_distribution_entropy=float(rng.random()), searched_value=float(np.float32(rng.standard_normal())),
                          predicted_value=np.array([rng.standard_normal()], np.float32))
            masks.append(m); tps.append(int(rng.integers(1, 3)) if ragged else -1)
            extra.append(ex)
            if sampled:
                out[e]["root_sampled_actions"] = ex
        nxt = rng.random((N,) + FRAME).astype(np.float32)
        rew = rng.standard_normal(N).astype(np.float32)
        done = rng.random(N) < 0.07
        fresh = rng.random((N,) + FRAME).astype(np.float32)   # first observation of the next episode (used where done)
        # (a) reference, per env (muzero_collector.py:606-692)
        for e in range(N):
            o = out[e]
            if sampled:
                segs[e].store_search_stats(o["visit_count_distributions"], o["searched_value"], extra[e])
            elif gumbel:
                segs[e].store_search_stats(o["visit_count_distributions"], o["searched_value"], improved_policy=extra[e])
            else:
                segs[e].store_search_stats(o["visit_count_distributions"], o["searched_value"])
            segs[e].append(o["action"], nxt[e], rew[e], masks[e], tps[e], t)
            if use_pri:
                pred_l[e].append(o["predicted_value"][0]); search_l[e].append(o["searched_value"])
            window[e] = window[e][1:] + [nxt[e]]
            if segs[e].is_full():
                if last[e] is not None:
                    pad_and_save(e, done[e])
                pri = priorities(e)
                pred_l[e], search_l[e] = [], []
                last[e], last_pri[e] = segs[e], pri
                segs[e] = GS(None, game_segment_length=L, config=cfg)
                segs[e].reset(window[e])
            if done[e]:
                if last[e] is not None:
                    pad_and_save(e, done[e])
                pri = priorities(e)
                segs[e].valid_transition_count = min(len(segs[e].action_segment), L)
                segs[e].game_segment_to_array()
                if len(segs[e].reward_segment) > 0:
                    pool.append((segs[e], pri, done[e]))
                segs[e] = GS(None, game_segment_length=L, config=cfg)
                window[e] = [fresh[e]] * STACK
                segs[e].reset(window[e])
                last[e], last_pri[e] = None, None
                pred_l[e], search_l[e] = [], []
        # (b) rows -> vectorised batch -> rollover
        rows = shard.pack_rows(out, masks, tps, AW, timestep=[t] * N, extra_key="root_sampled_actions" if sampled else None)
        batch.store_search_stats_rows(rows, improved_policy=np.stack(extra) if gumbel else None)   # sampled actions: from the row's extra block
        batch.append(nxt, rew)
        batch.rollover(done, reset_observations=fresh)
        for e in range(N):   # the collector's observation window, built on demand
            assert np.array_equal(batch.window(e), np.stack(window[e]).reshape(batch.window(e).shape)), (t, e)
    assert len(pool) > 12 and len(batch.pool) == len(pool)
    segs_b, meta = batch.drain_pool()
    assert batch.pool == [] and all(m["unroll_plus_td_steps"] == UNROLL + TD for m in meta)
    for k, ((rs, rp, rd), mine, m) in enumerate(zip(pool, segs_b, meta)):
        assert bool(rd) == m["done"], k
        assert (rp is None) == (m["priorities"] is None)
        if rp is not None:
            np.testing.assert_allclose(m["priorities"], rp, rtol=1e-6, atol=0)
        assert mine["valid_transition_count"] == rs.valid_transition_count
        assert np.array_equal(mine["obs_segment"], rs.obs_segment), k
        assert np.array_equal(mine["action_segment"], np.asarray(rs.action_segment)), k
        assert mine["action_segment"].shape == np.asarray(rs.action_segment).shape
        assert np.array_equal(mine["reward_segment"], np.asarray(rs.reward_segment, np.float32)), k
        assert np.array_equal(mine["action_mask_segment"], rs.action_mask_segment) and np.array_equal(mine["to_play_segment"], rs.to_play_segment)
        assert np.array_equal(mine["timestep_segment"], rs.timestep_segment)
        assert np.array_equal(mine["root_value_segment"], np.asarray(rs.root_value_segment, np.float32)), k
        theirs = rs.child_visit_segment
        assert len(mine["child_visit_segment"]) == len(theirs), k
        for a, b in zip(mine["child_visit_segment"], theirs):
            np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=2e-7, atol=1e-9)
        if gumbel:
            assert np.array_equal(mine["improved_policy_probs"], np.asarray(rs.improved_policy_probs, np.float32)), k
        if sampled:
            assert np.array_equal(mine["root_sampled_actions"], np.asarray(rs.root_sampled_actions, np.float32)), k


def test_partial_reset_keeps_the_other_envs_frames():
    """reset(env_ids=[2]) used to drop every frame appended with env_ids=None"""
    rng = np.random.default_rng(0)
    b = GameSegmentBatch(3, 2, 8, (1, 2, 2), frame_stack_num=1)
    init = rng.random((3, 1, 1, 2, 2)).astype(np.float32)
    b.reset(init)
    f1, f2 = rng.random((3, 1, 2, 2)).astype(np.float32), rng.random((3, 1, 2, 2)).astype(np.float32)
    rows = np.zeros((3, shard.HEADER + 4), np.float32)
    rows[:, shard.F_N_LEGAL] = 2
    for f in (f1, f2):
        b.store_search_stats_rows(rows)
        b.append(f, np.zeros(3))
    b.reset(rng.random((1, 1, 1, 2, 2)).astype(np.float32), env_ids=[2])
    for e in (0, 1):
        obs = b.to_arrays(e)["obs_segment"]
        assert np.array_equal(obs, np.stack([init[e, 0], f1[e], f2[e]]))
    assert b.to_arrays(2)["obs_segment"].shape[0] == 1 and b.len.tolist() == [2, 2, 0]
    f3 = rng.random((1, 1, 2, 2)).astype(np.float32)
    b.store_search_stats_rows(rows[:1], env_ids=[2])
    b.append(f3, [0.0], env_ids=[2])
    assert np.array_equal(b.to_arrays(2)["obs_segment"][1], f3[0]) and np.array_equal(b.to_arrays(0)["obs_segment"][2], f2[0])
    assert np.array_equal(b.window(2)[0], f3[0]) and np.array_equal(b.window(1)[0], f2[1])
