"""MuZeroVectorCollector.collect -- the collect loop of lzero/worker/muzero_collector.py:416-760 over a vectorised
environment -- against that loop restated per env on the reference's OWN GameSegment class (imported from /root/reference; the
collector module itself imports DI-engine): a stub environment with random episode ends and a stub policy with random search
statistics are run through the vectorised collector once, the transcript (policy rows, observations, rewards, dones, reset
observations) is replayed through the per-env logic, and the pooled segments, their priorities / done flags, the stacked observations
the policy was given, the ready-env bookkeeping and the episode statistics must agree."""
import numpy as np
import pytest

import ref_loader
from lightzero_amd import shard
from lightzero_amd.worker import MuZeroVectorCollector

N, A, L, STACK, FRAME, UNROLL, TD = 4, 5, 5, 3, (1, 3, 3), 2, 2


class StubEnv:
    def __init__(self, rng, ragged):
        self.env_num, self.rng, self.ragged, self.steps = N, rng, ragged, []
        self.t = np.zeros(N, np.int64)

    def _obs(self):
        m = np.ones((N, A), np.float32)
        if self.ragged:
            m = (self.rng.random((N, A)) < 0.6).astype(np.float32)
            m[np.arange(N), self.rng.integers(0, A, N)] = 1
        return dict(observation=self.rng.random((N,) + FRAME).astype(np.float32), action_mask=m,
                    to_play=self.rng.integers(1, 3, N) if self.ragged else np.full(N, -1), timestep=self.t.copy())

    def reset(self):
        self.t[:] = 0
        self.first = self._obs()
        return self.first

    def step(self, actions, active):
        self.t += 1
        obs = self._obs()
        reward = self.rng.standard_normal(N).astype(np.float32)
        done = (self.rng.random(N) < 0.07) & active
        self.t[done] = 0
        reset_obs = self._obs()
        info = dict(reset_obs=reset_obs, eval_episode_return=self.rng.standard_normal(N))
        self.steps.append(dict(actions=actions.copy(), active=active.copy(), obs=obs, reward=reward, done=done.copy(), info=info))
        return obs, reward, done, info


class StubPolicy:
    def __init__(self, rng):
        self.rng, self.calls = rng, []

    def forward_collect_rows(self, data, action_mask, rows_out, temperature=1, to_play=(-1,), timestep=None, frame_floats=None, epsilon=0.0):
        out = {}
        for e in range(N):
            legal = np.nonzero(action_mask[e])[0]
            out[e] = dict(action=int(legal[self.rng.integers(0, len(legal))]), visit_count_distributions=self.rng.integers(0, 20, len(legal)).tolist(),
                          visit_count_distribution_entropy=float(self.rng.random()), searched_value=float(np.float32(self.rng.standard_normal())),
                          predicted_value=np.array([self.rng.standard_normal()], np.float32))
        rows = shard.pack_rows(out, [action_mask[e] for e in range(N)], list(to_play), A, timestep=list(timestep))
        self.calls.append(dict(out=out, data=np.array(data), mask=np.array(action_mask), to_play=list(to_play), timestep=list(timestep)))
        return rows


def _cfg(use_pri, ignore_done):
    from easydict import EasyDict
    return EasyDict(dict(num_unroll_steps=UNROLL, td_steps=TD, discount_factor=0.997, gray_scale=False, transform2string=False,
                         sampled_algo=False, gumbel_algo=False, use_ture_chance_label_in_chance_encoder=False, game_segment_length=L,
                         use_priority=use_pri, use_max_priority_for_new_data=False, ignore_done=ignore_done,
                         model=dict(frame_stack_num=STACK, action_space_size=A, image_channel=1, observation_shape=(STACK, 3, 3))))


@pytest.mark.parametrize("mode,n_episode", [("plain", 6), ("ragged_priority", 11), ("ignore_done", 7)])
def test_collect_equals_the_reference_loop_on_reference_segments(mode, n_episode):
    ref = ref_loader.load()
    if ref is None:
        pytest.skip("/root/reference not present")
    ragged = use_pri = mode == "ragged_priority"
    cfg = _cfg(use_pri, mode == "ignore_done")
    seed = {"plain": 11, "ragged_priority": 12, "ignore_done": 13}[mode]
    env, pol = StubEnv(np.random.default_rng(seed), ragged), StubPolicy(np.random.default_rng(seed + 100))
    col = MuZeroVectorCollector(env, pol, cfg, device=None)
    segs_v, meta_v = col.collect(n_episode=n_episode)
    _check_against_reference_loop(ref, mode, cfg, env, pol, n_episode, segs_v, meta_v, col.episode_info)
    assert col.total_episode_count == len(col.episode_info) and col.total_envstep_count == sum(int(s["active"].sum()) for s in env.steps)


@pytest.mark.parametrize("groups", [2, 3])
def test_pipelined_env_groups_each_equal_the_reference_loop(groups):
    """env groups: the collector interleaves the groups (one forward enqueued per group, the host steps one group's envs while the device
    searches for another); every group's transcript still replays exactly through the reference loop, and the engine-facing calls never overlap"""
    ref = ref_loader.load()
    if ref is None:
        pytest.skip("/root/reference not present")
    import threading
    cfg = _cfg(True, False)
    envs = [StubEnv(np.random.default_rng(40 + g), True) for g in range(groups)]
    pols = [StubPolicy(np.random.default_rng(140 + g)) for g in range(groups)]
    in_flight, overlaps, threads = [0], [0], set()
    lock = threading.Lock()
    for p in pols:   # the forwards run one at a time
        inner = p.forward_collect_rows

        def spy(*a, _inner=inner, **kw):
            with lock:
                in_flight[0] += 1
                overlaps[0] += in_flight[0] > 1
                threads.add(threading.current_thread().name)
            try:
                return _inner(*a, **kw)
            finally:
                with lock:
                    in_flight[0] -= 1
        p.forward_collect_rows = spy
    col = MuZeroVectorCollector(envs, pols, cfg, device=None)
    n_episode = 9 * groups + 1
    segs_v, meta_v = col.collect(n_episode=n_episode)
    # round 5: no worker thread any more -- a policy with a split forward keeps one forward ENQUEUED per group (forward_collect_rows_begin
    # returns without waiting for the device); a policy without one (this stub) runs whole where the collector waits for the group
    assert overlaps[0] == 0 and threads == {threading.current_thread().name}
    share = [n_episode // groups + (1 if g < n_episode % groups else 0) for g in range(groups)]
    assert len(col.group_results) == groups and sum(len(r[0]) for r in col.group_results) == len(segs_v)
    for g in range(groups):
        sv, mv, info = col.group_results[g]
        _check_against_reference_loop(ref, "ragged_priority", cfg, envs[g], pols[g], share[g], sv, mv, info)
    assert col.total_episode_count == sum(len(r[2]) for r in col.group_results) >= n_episode


def _check_against_reference_loop(ref, mode, cfg, env, pol, n_episode, segs_v, meta_v, episode_info_v):
    GS = ref.game_segment.GameSegment
    ragged = use_pri = mode == "ragged_priority"
    assert len(env.steps) == len(pol.calls) > 10
    # ---- the reference loop (muzero_collector.py:470-735) over the same transcript
    init = env.first
    window = [[init["observation"][e]] * STACK for e in range(N)]
    segs = [GS(None, game_segment_length=L, config=cfg) for _ in range(N)]
    for e in range(N):
        segs[e].reset(window[e])
    mask = [init["action_mask"][e] for e in range(N)]
    tp = [int(init["to_play"][e]) for e in range(N)]
    ts = [int(init["timestep"][e]) for e in range(N)]
    last, last_pri, pool = [None] * N, [None] * N, []
    pred_l, search_l = [[] for _ in range(N)], [[] for _ in range(N)]
    eps_steps, ent = np.zeros(N), np.zeros(N)
    ready, remain, collected, episode_info = set(), n_episode, 0, []
    finished_now = set(range(N))   # "new available" envs of the next iteration (:513-516)
    p = UNROLL + TD

    def priorities(e):
        if not use_pri:
            return None
        return np.abs(np.asarray(pred_l[e], np.float32) - np.asarray(search_l[e], np.float32)) + np.float32(1e-6)

    def pad_and_save(e, flag):
        g = segs[e]
        last[e].valid_transition_count = min(len(last[e].action_segment), L)
        last[e].pad_over(g.obs_segment[STACK:STACK + p], g.reward_segment[:p - 1], g.action_segment[:p], g.root_value_segment[:p], g.child_visit_segment[:p])
        last[e].game_segment_to_array()
        pool.append((last[e], last_pri[e], flag))
        last[e], last_pri[e] = None, None

    for k, (stp, call) in enumerate(zip(env.steps, pol.calls)):
        new = sorted(finished_now.difference(ready))
        ready.update(new[:remain])
        remain -= min(len(new), remain)
        finished_now = set()
        assert sorted(ready) == np.nonzero(stp["active"])[0].tolist(), k
        for e in sorted(ready):   # what the policy saw for the ready envs: stacked observation, mask, to_play, timestep
            assert np.array_equal(call["data"][e], np.concatenate(segs[e].get_obs(), 0)), (k, e)
            assert np.array_equal(call["mask"][e], mask[e]) and call["to_play"][e] == tp[e] and call["timestep"][e] == ts[e]
        for e in sorted(ready):
            o = call["out"][e]
            assert stp["actions"][e] == o["action"]
            segs[e].store_search_stats(o["visit_count_distributions"], o["searched_value"])
            segs[e].append(o["action"], stp["obs"]["observation"][e], stp["reward"][e], mask[e], tp[e], ts[e])
            mask[e], tp[e], ts[e] = stp["obs"]["action_mask"][e], int(stp["obs"]["to_play"][e]), int(stp["obs"]["timestep"][e])
            done = bool(stp["done"][e])
            flag = done if mode != "ignore_done" else False
            ent[e] += o["visit_count_distribution_entropy"]; eps_steps[e] += 1
            if use_pri:
                pred_l[e].append(o["predicted_value"][0]); search_l[e].append(o["searched_value"])
            window[e] = window[e][1:] + [stp["obs"]["observation"][e]]
            if segs[e].is_full():
                if last[e] is not None:
                    pad_and_save(e, flag)
                pri = priorities(e)
                pred_l[e], search_l[e] = [], []
                last[e], last_pri[e] = segs[e], pri
                segs[e] = GS(None, game_segment_length=L, config=cfg)
                segs[e].reset(window[e])
            if done:
                collected += 1
                episode_info.append(dict(reward=float(stp["info"]["eval_episode_return"][e]), step=int(eps_steps[e]), visit_entropy=float(ent[e] / eps_steps[e])))
                if last[e] is not None:
                    pad_and_save(e, flag)
                pri = priorities(e)
                segs[e].valid_transition_count = min(len(segs[e].action_segment), L)
                segs[e].game_segment_to_array()
                if len(segs[e].reward_segment) > 0:
                    pool.append((segs[e], pri, flag))
                ro = stp["info"]["reset_obs"]
                mask[e], tp[e], ts[e] = ro["action_mask"][e], int(ro["to_play"][e]), int(ro["timestep"][e])
                segs[e] = GS(None, game_segment_length=L, config=cfg)
                window[e] = [ro["observation"][e]] * STACK
                segs[e].reset(window[e])
                last[e], last_pri[e] = None, None
                pred_l[e], search_l[e] = [], []
                eps_steps[e], ent[e] = 0, 0
                ready.remove(e)
                finished_now.add(e)
        if collected >= n_episode:
            assert k == len(env.steps) - 1   # the collector stopped exactly here
            break
    assert collected >= n_episode
    assert len(episode_info_v) == len(episode_info) == collected
    for a, b in zip(episode_info_v, episode_info):
        assert a["reward"] == b["reward"] and a["step"] == b["step"] and abs(a["visit_entropy"] - b["visit_entropy"]) < 1e-6
    assert len(pool) == len(segs_v) > n_episode - 1
    for k, ((rs, rp, rd), mine, m) in enumerate(zip(pool, segs_v, meta_v)):
        assert bool(rd) == m["done"] and m["unroll_plus_td_steps"] == p, k
        assert (rp is None) == (m["priorities"] is None)
        if rp is not None:
            np.testing.assert_allclose(m["priorities"], rp, rtol=1e-6, atol=0)
        assert mine["valid_transition_count"] == rs.valid_transition_count
        assert np.array_equal(mine["obs_segment"], rs.obs_segment), k
        assert np.array_equal(mine["action_segment"], rs.action_segment), k
        assert np.array_equal(mine["reward_segment"], np.asarray(rs.reward_segment, np.float32)), k
        assert np.array_equal(mine["action_mask_segment"], rs.action_mask_segment) and np.array_equal(mine["to_play_segment"], rs.to_play_segment)
        assert np.array_equal(mine["timestep_segment"], rs.timestep_segment)
        assert np.array_equal(mine["root_value_segment"], np.asarray(rs.root_value_segment, np.float32)), k
        assert len(mine["child_visit_segment"]) == len(rs.child_visit_segment), k
        for a, b in zip(mine["child_visit_segment"], rs.child_visit_segment):
            np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=2e-7, atol=1e-9)
