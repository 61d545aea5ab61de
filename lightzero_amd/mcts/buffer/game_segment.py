"""Vectorised counterpart of lzero/mcts/buffer/game_segment.py::GameSegment for a whole batch of environments: what
``MuZeroCollector.collect`` does per env and step with Python lists (muzero_collector.py:588-620:
``store_search_stats`` game_segment.py:241-263 + ``append`` :158-182) is two array writes for all envs here, fed by the packed
env-step rows the engine writes on the device (``lz_roots_collect_rows`` / lightzero_amd.shard.pack_rows).

Per env it holds exactly the lists of the reference class as pre-allocated arrays (``obs_segment`` incl. the ``frame_stack_num``
initial frames of ``reset``, ``action_segment``, ``reward_segment``, ``child_visit_segment``, ``root_value_segment``,
``action_mask_segment``, ``to_play_segment``, ``timestep_segment``); ``to_arrays(env)`` returns what
``game_segment_to_array`` (:265-338) leaves in those attributes.  Replay buffers, targets and ``pad_over`` stay out of scope."""
import numpy as np

from ... import shard


class GameSegmentBatch(object):
    def __init__(self, n_env, action_space_size, game_segment_length, frame_shape, frame_stack_num=1, extra=0):
        """``extra``: room for the ``num_unroll_steps + td_steps`` entries ``pad_over`` may add in the reference (unused here)."""
        self.n_env, self.A, self.L = int(n_env), int(action_space_size), int(game_segment_length)
        self.frame_shape, self.stack = tuple(frame_shape), int(frame_stack_num)
        cap = self.L + int(extra)
        # observation frames are kept the way the reference keeps them -- by reference (its obs_segment is a list of the arrays
        # the environment returned): one entry per append call = (env ids or None, [n, *frame_shape] array, per-env positions)
        self._init_obs = np.zeros((self.n_env, self.stack) + self.frame_shape, np.float32)
        self._frames = []
        self.action = np.zeros((self.n_env, cap), np.int64)
        self.reward = np.zeros((self.n_env, cap), np.float32)
        self.child_visits = np.zeros((self.n_env, cap, self.A), np.float32)
        self.n_legal = np.zeros((self.n_env, cap), np.int64)
        self.root_value = np.zeros((self.n_env, cap), np.float32)
        self.action_mask = np.zeros((self.n_env, cap, self.A), np.float32)
        self.to_play = np.zeros((self.n_env, cap), np.int64)
        self.timestep = np.zeros((self.n_env, cap), np.int64)
        self.predicted_value = np.zeros((self.n_env, cap), np.float32)
        self.entropy = np.zeros((self.n_env, cap), np.float32)
        self.len = np.zeros(self.n_env, np.int64)       # transitions appended so far (= len(action_segment))
        self._stats = np.zeros(self.n_env, np.int64)    # search statistics stored so far (= len(root_value_segment))

    def _ids(self, env_ids):
        return np.arange(self.n_env) if env_ids is None else np.asarray(env_ids, np.int64)

    def reset(self, init_observations, env_ids=None):
        """GameSegment.reset (game_segment.py:340-368): start a segment from the ``frame_stack_num`` previous frames,
        ``init_observations`` [n, frame_stack_num, *frame_shape]"""
        ids = self._ids(env_ids)
        init = np.asarray(init_observations, np.float32).reshape((len(ids), self.stack) + self.frame_shape)
        self._init_obs[ids] = init
        if env_ids is None:
            self._frames = []
        else:  # the appended frames of these envs belong to their previous segment
            drop = set(int(i) for i in ids)
            self._frames = [(fi, fr, pos) for fi, fr, pos in self._frames if fi is not None and not drop.intersection(fi.tolist())]
        self.len[ids] = 0
        self._stats[ids] = 0

    def store_search_stats_rows(self, rows, env_ids=None):
        """store_search_stats (:241-263) for every env of ``rows`` ([n, >= 8 + 2A] env-step rows, frames not needed): child
        visits / sum and root value; also keeps the decision-time fields ``append`` will take (action, mask, to_play, timestep)."""
        ids = self._ids(env_ids)
        rows = np.asarray(rows)
        t = self._stats[ids]
        A, H = self.A, shard.HEADER
        self.child_visits[ids, t] = rows[:, H:H + A]
        self.n_legal[ids, t] = rows[:, shard.F_N_LEGAL].astype(np.int64)
        self.root_value[ids, t] = rows[:, shard.F_ROOT_VALUE]
        self.predicted_value[ids, t] = rows[:, shard.F_PRED_VALUE]
        self.entropy[ids, t] = rows[:, shard.F_ENTROPY]
        self.action[ids, t] = rows[:, shard.F_ACTION].astype(np.int64)
        self.action_mask[ids, t] = rows[:, H + A:H + 2 * A]
        self.to_play[ids, t] = rows[:, shard.F_TO_PLAY].astype(np.int64)
        self.timestep[ids, t] = rows[:, shard.F_TIMESTEP].astype(np.int64)
        self._stats[ids] = t + 1

    def append(self, next_observations, rewards, env_ids=None):
        """the environment-side half of GameSegment.append (:158-182): o_{t+1} and r_t of the transition whose decision-time
        fields came with ``store_search_stats_rows``"""
        ids = self._ids(env_ids)
        t = self.len[ids]
        frames = np.asarray(next_observations, np.float32).reshape((len(ids),) + self.frame_shape)  # a view when already float32
        self._frames.append((None if env_ids is None else ids.copy(), frames, t.copy()))
        self.reward[ids, t] = np.asarray(rewards, np.float32)
        self.len[ids] = t + 1

    def is_full(self):
        """GameSegment.is_full (:370-377), per env"""
        return self.len >= self.L

    def to_arrays(self, env):
        """game_segment_to_array (:265-338) for one env: the arrays its attributes hold afterwards"""
        n = int(self.len[env])
        ns = int(self._stats[env])
        nl = self.n_legal[env, :ns]
        if ns and (nl == nl[0]).all():
            child = self.child_visits[env, :ns, :int(nl[0])].copy()
        else:  # variable action spaces (board games): dtype=object like the reference (:316-321)
            child = np.empty(ns, dtype=object)
            for k in range(ns):
                child[k] = self.child_visits[env, k, :int(nl[k])].tolist()
        obs = np.zeros((self.stack + n,) + self.frame_shape, np.float32)
        obs[:self.stack] = self._init_obs[env]
        for fi, fr, pos in self._frames:
            if fi is None:
                obs[self.stack + int(pos[env])] = fr[env]
            else:
                k = np.nonzero(fi == env)[0]
                if k.size:
                    obs[self.stack + int(pos[k[0]])] = fr[k[0]]
        return dict(obs_segment=obs, action_segment=self.action[env, :n].copy(),
                    reward_segment=self.reward[env, :n].copy(), child_visit_segment=child,
                    root_value_segment=self.root_value[env, :ns].copy(), action_mask_segment=self.action_mask[env, :n].copy(),
                    to_play_segment=self.to_play[env, :n].copy(), timestep_segment=self.timestep[env, :n].copy())
