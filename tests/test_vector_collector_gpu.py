"""GPU: MuZeroVectorCollector with the real EfficientZeroPolicy on the engine model -- device-resident frame stack (only the newest
frame of every env is uploaded per step), rows written by the device, segments pooled; checked: every pooled segment is well
formed (lengths, padding, child visits normalised over the legal actions of the mask the policy saw), the frames the device wrote
into the rows are the newest frames of the stack the collector kept, episode / env-step counts."""
import numpy as np
import pytest
import torch

from lightzero_amd import shard

pytestmark = pytest.mark.gpu
N, A, L, STACK, UNROLL, TD = 24, 6, 6, 4, 3, 2


class FrameEnv:
    """synthetic Atari-shaped env: random 1x96x96 frames, ragged masks, random episode ends"""
    def __init__(self, seed):
        self.env_num, self.rng, self.t = N, np.random.default_rng(seed), np.zeros(N, np.int64)
        self.frames_seen = []

    def _obs(self):
        m = (self.rng.random((N, A)) < 0.7).astype(np.float32)
        m[np.arange(N), self.rng.integers(0, A, N)] = 1
        return dict(observation=self.rng.random((N, 1, 96, 96)).astype(np.float32), action_mask=m, to_play=np.full(N, -1), timestep=self.t.copy())

    def reset(self):
        self.t[:] = 0
        return self._obs()

    def step(self, actions, active):
        self.t += 1
        obs = self._obs()
        done = (self.rng.random(N) < 0.08) & active
        self.t[done] = 0
        return obs, self.rng.standard_normal(N).astype(np.float32), done, dict(reset_obs=self._obs(), eval_episode_return=self.rng.standard_normal(N))


def test_collect_with_the_engine_policy():
    from oracle import torch_models as tm
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    from lightzero_amd.policy.efficientzero import EfficientZeroPolicy
    from lightzero_amd.worker import MuZeroVectorCollector
    sd = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A), seed=3).state_dict()
    model = EfficientZeroModel(action_space_size=A).load_state_dict(sd)
    cfg = dict(num_simulations=12, pb_c_base=19652, pb_c_init=1.25, discount_factor=0.997, value_delta_max=0.01, lstm_horizon_len=5,
               root_noise_weight=0.25, root_dirichlet_alpha=0.3, game_segment_length=L, num_unroll_steps=UNROLL, td_steps=TD,
               use_priority=True, model=dict(frame_stack_num=STACK, action_space_size=A))
    pol = EfficientZeroPolicy(cfg, model)
    seen = []
    inner = pol.forward_collect_rows

    def spy(data, mask, rows_out, **kw):   # what the policy was given and what the device wrote
        hdr = inner(data, mask, rows_out, **kw)
        seen.append(dict(newest=data[:, -1].cpu().numpy().copy(), frame=rows_out[:, shard.HEADER + 2 * A:].cpu().numpy().copy(), mask=np.array(mask), hdr=hdr.copy()))
        return hdr
    pol.forward_collect_rows = spy
    env = FrameEnv(5)
    col = MuZeroVectorCollector(env, pol, cfg, device="cuda")
    n_episode = N + 6
    segs, meta = col.collect(n_episode=n_episode, policy_kwargs=dict(temperature=1.0, epsilon=0.0))
    assert col.total_episode_count >= n_episode and len(col.episode_info) == col.total_episode_count
    assert len(segs) == len(meta) >= n_episode and col.total_envstep_count > 0
    for s_ in seen:   # the row's frame block is the newest frame of the stacked observation
        assert np.array_equal(s_["frame"].reshape(N, 96, 96), s_["newest"])
        assert np.array_equal(s_["hdr"][:, shard.HEADER + A:shard.HEADER + 2 * A], s_["mask"])
        assert all(s_["mask"][i, int(s_["hdr"][i, shard.F_ACTION])] == 1 for i in range(N))
    for seg, m in zip(segs, meta):
        n = len(seg["action_segment"])
        assert 0 < seg["valid_transition_count"] <= L and n <= L + UNROLL + TD
        assert seg["obs_segment"].shape[0] == STACK + n and seg["obs_segment"].shape[1:] == (1, 96, 96)
        assert len(seg["root_value_segment"]) == len(seg["child_visit_segment"]) == n
        for cv, mk in zip(seg["child_visit_segment"][:seg["valid_transition_count"]], seg["action_mask_segment"]):
            assert len(cv) == int(mk.sum()) and abs(float(np.sum(cv)) - 1.0) < 1e-5
        assert m["unroll_plus_td_steps"] == UNROLL + TD
        assert m["priorities"] is not None and len(m["priorities"]) == seg["valid_transition_count"] and (m["priorities"] > 0).all()
