"""the packed env-step row and the vectorised GameSegment batch against the reference's OWN per-env code.
For a batch of envs and a few steps, the same policy output dicts drive
  (a) the reference path, per env and step as MuZeroCollector.collect does (muzero_collector.py:588-620):
      GameSegment.store_search_stats (game_segment.py:241-263) + GameSegment.append (:158-182), then game_segment_to_array
      (:265-338) -- the reference class itself, imported from /root/reference (tests/ref_loader.py; skipped where it is absent),
  (b) rows = shard.pack_rows(...) -> GameSegmentBatch.store_search_stats_rows / append -> to_arrays,
and every array must be equal.  The round trip through the all-gather representation (unpack_rows) is checked too."""
import numpy as np
import pytest

import ref_loader
from lightzero_amd import shard
from lightzero_amd.mcts.buffer.game_segment import GameSegmentBatch

N, A, T, STACK, FRAME = 5, 6, 7, 4, (1, 8, 8)


def _policy_outputs(rng, ragged):
    """T steps of per-env policy output dicts (the efficientzero.py:636-643 contract) + masks / to_play / env returns"""
    steps = []
    for t in range(T):
        out, masks, tps = {}, [], []
        for e in range(N):
            m = np.ones(A, np.float32)
            if ragged:
                m = (rng.random(A) < 0.6).astype(np.float32)
                m[rng.integers(0, A)] = 1
            legal = np.nonzero(m)[0]
            visits = rng.integers(0, 20, size=len(legal)).tolist()
            out[e] = dict(action=int(legal[rng.integers(0, len(legal))]), visit_count_distributions=visits,
                          visit_count_distribution_entropy=float(rng.random()), searched_value=float(rng.standard_normal()),
                          predicted_value=np.array([rng.standard_normal()], np.float32), predicted_policy_logits=rng.standard_normal(A).tolist())
            masks.append(m)
            tps.append(int(rng.integers(1, 3)) if ragged else -1)
        steps.append(dict(out=out, masks=masks, to_play=tps, next_obs=rng.random((N,) + FRAME).astype(np.float32),
                          reward=rng.standard_normal(N).astype(np.float32), timestep=[t] * N))
    return steps


@pytest.mark.parametrize("ragged", [False, True])
def test_rows_and_segment_batch_equal_reference_game_segment(ragged):
    ref = ref_loader.load()
    if ref is None:
        pytest.skip("/root/reference not present")
    from easydict import EasyDict
    rng = np.random.default_rng(7 + int(ragged))
    cfg = EasyDict(dict(num_unroll_steps=5, td_steps=5, discount_factor=0.997, gray_scale=False, transform2string=False,
                        sampled_algo=False, gumbel_algo=False, use_ture_chance_label_in_chance_encoder=False,
                        model=dict(frame_stack_num=STACK, action_space_size=A, image_channel=1, observation_shape=(STACK, 8, 8))))
    init = rng.random((N, STACK) + FRAME).astype(np.float32)
    segs = [ref.game_segment.GameSegment(None, game_segment_length=T, config=cfg) for _ in range(N)]
    for e in range(N):
        segs[e].reset([init[e, k] for k in range(STACK)])
    batch = GameSegmentBatch(N, A, T, FRAME, frame_stack_num=STACK)
    batch.reset(init)
    for st in _policy_outputs(rng, ragged):
        # (a) the reference, per env (muzero_collector.py:606-619)
        for e in range(N):
            o = st["out"][e]
            segs[e].store_search_stats(o["visit_count_distributions"], o["searched_value"])
            segs[e].append(o["action"], st["next_obs"][e], st["reward"][e], st["masks"][e], st["to_play"][e], st["timestep"][e])
        # (b) packed rows -> vectorised batch; the rows also survive the all-gather representation
        rows = shard.pack_rows(st["out"], st["masks"], st["to_play"], A, timestep=st["timestep"])
        cols = shard.unpack_rows(rows, A)
        assert np.array_equal(cols["action"], [st["out"][e]["action"] for e in range(N)])
        batch.store_search_stats_rows(rows)
        batch.append(st["next_obs"], st["reward"])
    assert batch.is_full().all() and all(s.is_full() for s in segs)
    for e in range(N):
        segs[e].game_segment_to_array()
        mine = batch.to_arrays(e)
        assert np.array_equal(mine["obs_segment"], segs[e].obs_segment)
        assert np.array_equal(mine["action_segment"], segs[e].action_segment)
        assert np.array_equal(mine["reward_segment"], segs[e].reward_segment)
        assert np.array_equal(mine["action_mask_segment"], segs[e].action_mask_segment)
        assert np.array_equal(mine["to_play_segment"], segs[e].to_play_segment)
        assert np.array_equal(mine["timestep_segment"], segs[e].timestep_segment)
        np.testing.assert_allclose(mine["root_value_segment"], np.asarray(segs[e].root_value_segment, np.float32), rtol=0, atol=0)
        theirs = segs[e].child_visit_segment
        assert mine["child_visit_segment"].dtype == theirs.dtype or theirs.dtype != object
        for k in range(T):  # the reference divides Python ints by a Python float sum (float64); the row is float32
            np.testing.assert_allclose(np.asarray(mine["child_visit_segment"][k], np.float64), np.asarray(theirs[k], np.float64), rtol=2e-7, atol=1e-9)


def test_row_layout_constants():
    assert shard.row_width(6, 96 * 96) == 8 + 12 + 9216          # 36.9 KB per env-step in float32 (~37 KB)
    out = {0: dict(action=2, visit_count_distributions=[0, 0], searched_value=0.0, predicted_value=0.0)}
    rows = shard.pack_rows(out, [np.array([0, 0, 1, 1], np.float32)], -1, 4)
    assert np.isfinite(rows).all() and rows[0, shard.HEADER:shard.HEADER + 2].tolist() == [0.0, 0.0]  # sum of visits 0 -> 1e-6 (game_segment.py:244)
