#!/usr/bin/env python
"""cProfile of the single-group MuZeroVectorCollector.collect loop at BASELINE configs[1] (256 envs, synthetic env): where the host
time of an env-step goes beside the 3.3 ms search.   python tools/prof_collector.py"""
import cProfile, io, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightzero_amd.model.efficientzero_model import EfficientZeroModel
from lightzero_amd.model.synthetic import efficientzero_state_dict
from lightzero_amd.policy.efficientzero import EfficientZeroPolicy
from lightzero_amd.worker import MuZeroVectorCollector
B, A = 256, 6
model = EfficientZeroModel(action_space_size=A).load_state_dict(efficientzero_state_dict(seed=0, action_space_size=A))


class _Env:
    def __init__(self):
        self.env_num, self.rng, self.k = B, np.random.default_rng(0), 0
        self.pool = [np.random.default_rng(i).random((B, 1, 96, 96), dtype=np.float32) for i in range(4)]
        self.mask, self.tp = np.ones((B, A), np.float32), np.full(B, -1)

    def _obs(self):
        self.k += 1
        return dict(observation=self.pool[self.k % 4], action_mask=self.mask, to_play=self.tp)

    def reset(self):
        return self._obs()

    def step(self, actions, active):
        done = (self.rng.random(B) < 1.0 / 150) & active
        return self._obs(), np.zeros(B, np.float32), done, dict(reset_obs=self._obs(), eval_episode_return=np.zeros(B))


ccfg = dict(num_simulations=50, discount_factor=0.997, lstm_horizon_len=5, game_segment_length=400, num_unroll_steps=5, td_steps=5,
            model=dict(frame_stack_num=4, action_space_size=A))
col = MuZeroVectorCollector(_Env(), EfficientZeroPolicy(ccfg, model), ccfg, device="cuda")
col.collect(n_episode=B)
s0 = col.total_envstep_count
t0 = time.perf_counter()
col.collect(n_episode=B)
dt = time.perf_counter() - t0
n = col.total_envstep_count - s0
print("collect: %.0f env-steps/s (%.2f ms per 256-env step)" % (n / dt, dt / (n / B) * 1e3))
pr = cProfile.Profile(); pr.enable()
s0 = col.total_loop_steps
col.collect(n_episode=B)
pr.disable()
steps = col.total_loop_steps - s0
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(16)
print("loop steps profiled:", steps)
print(s.getvalue()[:4200])
