"""Vectorised counterpart of lzero/mcts/buffer/game_segment.py::GameSegment for a whole batch of environments, plus the segment
hand-over of the collector: what ``MuZeroCollector.collect`` does per env and step with Python lists (muzero_collector.py:588-620:
``store_search_stats`` game_segment.py:241-263 + ``append`` :158-182) is two array writes for all envs here, fed by the packed
env-step rows the engine writes on the device (``lz_roots_collect_rows`` / lightzero_amd.shard.pack_rows); ``rollover`` is the
collector's segment logic (muzero_collector.py:308-410, 649-692): a full segment waits for the first ``num_unroll_steps + td_steps``
entries of its successor (``pad_over``, game_segment.py:183-234), then goes to the pool with its priorities and done flag.

Per env it holds exactly the lists of the reference class as pre-allocated arrays (``obs_segment`` incl. the ``frame_stack_num``
initial frames of ``reset``, ``action_segment``, ``reward_segment``, ``child_visit_segment``, ``root_value_segment``,
``action_mask_segment``, ``to_play_segment``, ``timestep_segment``, and for the sampled / Gumbel families ``root_sampled_actions`` /
``improved_policy_probs``); ``to_arrays(env)`` returns what ``game_segment_to_array`` (:265-338) leaves in those attributes.
Replay buffers and target computation stay out of scope."""
import numpy as np

from ... import shard


class GameSegmentBatch(object):
    def __init__(self, n_env, action_space_size, game_segment_length, frame_shape, frame_stack_num=1, num_unroll_steps=5, td_steps=5,
                 sampled_actions_shape=None, improved_policy=False, use_priority=False, use_max_priority_for_new_data=False,
                 ignore_done=False, extra=0, continuous_action_space=False):
        """``sampled_actions_shape`` (K, D): Sampled EfficientZero -- the row's child visits are over the K sampled actions and every
        step also stores the root's sampled actions (game_segment.py:254-255).  ``improved_policy``: Gumbel MuZero -- every step stores
        the improved policy over the A actions (:257-258).  ``use_priority`` / ``use_max_priority_for_new_data``: muzero_collector.py:
        308-334.  ``continuous_action_space`` (with ``sampled_actions_shape``): the row's action word is the POSITION of the chosen action
        among the K sampled ones; what the segment stores is the action itself, a [D] vector (sampled_efficientzero.py:905-913) -- or, for a
        discrete sampled space, its first component as an int."""
        self.n_env, self.A, self.L = int(n_env), int(action_space_size), int(game_segment_length)
        self.frame_shape, self.stack = tuple(frame_shape), int(frame_stack_num)
        self.pad = int(num_unroll_steps) + int(td_steps)
        self.use_priority = bool(use_priority) and not bool(use_max_priority_for_new_data)
        self.ignore_done = bool(ignore_done)
        cap = self.L + int(extra)
        # observation frames are kept the way the reference keeps them -- by reference (its obs_segment is a list of the arrays the
        # environment returned): one entry per append call = (env ids or None, [n, *frame_shape] array, per-env positions, per-env
        # segment generation).  An entry belongs to an env's CURRENT segment while its generation tag equals gen[env]; ``reset`` of
        # some envs bumps their generation and leaves every other env's frames alone.
        self._init_obs = np.zeros((self.n_env, self.stack) + self.frame_shape, np.float32)
        self._frames = []
        self._base = 0                                   # entries dropped from the front of _frames so far
        self._first = np.zeros(self.n_env, np.int64)     # per env: absolute index of the first entry that can belong to its current segment
        self._gen = np.zeros(self.n_env, np.int64)
        self.continuous = bool(continuous_action_space) and sampled_actions_shape is not None
        self.action = (np.zeros((self.n_env, cap, int(sampled_actions_shape[1])), np.float32) if self.continuous
                       else np.zeros((self.n_env, cap), np.int64))
        self.reward = np.zeros((self.n_env, cap), np.float32)
        self.child_visits = np.zeros((self.n_env, cap, self.A), np.float32)
        self.n_legal = np.zeros((self.n_env, cap), np.int64)
        self.root_value = np.zeros((self.n_env, cap), np.float32)
        self.action_mask = np.zeros((self.n_env, cap, self.A), np.float32)
        self.to_play = np.zeros((self.n_env, cap), np.int64)
        self.timestep = np.zeros((self.n_env, cap), np.int64)
        self.predicted_value = np.zeros((self.n_env, cap), np.float32)
        self.entropy = np.zeros((self.n_env, cap), np.float32)
        self.sampled_shape = tuple(sampled_actions_shape) if sampled_actions_shape else None
        self.sampled_actions = np.zeros((self.n_env, cap) + self.sampled_shape, np.float32) if self.sampled_shape else None
        self.improved = np.zeros((self.n_env, cap, self.A), np.float32) if improved_policy else None
        self.len = np.zeros(self.n_env, np.int64)       # transitions appended so far (= len(action_segment))
        self._stats = np.zeros(self.n_env, np.int64)    # search statistics stored so far (= len(root_value_segment))
        self._last = [None] * self.n_env                # the previous, full segment of an env: waits for its padding
        self._last_pri = [None] * self.n_env
        self.pool = []                                  # (segment dict, priorities | None, done): muzero_collector.py game_segment_pool

    def _ids(self, env_ids):
        return np.arange(self.n_env) if env_ids is None else np.asarray(env_ids, np.int64)

    def reset(self, init_observations, env_ids=None):
        """GameSegment.reset (game_segment.py:340-368): start a segment from the ``frame_stack_num`` previous frames,
        ``init_observations`` [n, frame_stack_num, *frame_shape]"""
        ids = self._ids(env_ids)
        init = np.asarray(init_observations, np.float32).reshape((len(ids), self.stack) + self.frame_shape)
        self._init_obs[ids] = init
        self._gen[ids] += 1   # the frames appended so far belong to the previous segments of these envs -- and only of these
        self._first[ids] = self._base + len(self._frames)    # the new segments' frames are appended from here on
        drop = int(self._first.min()) - self._base            # entries in front of every env's current segment
        if drop > 64:
            del self._frames[:drop]
            self._base += drop
        self.len[ids] = 0
        self._stats[ids] = 0

    def store_search_stats_rows(self, rows, env_ids=None, sampled_actions=None, improved_policy=None):
        """store_search_stats (:241-263) for every env of ``rows`` ([n, >= 8 + 2A] env-step rows, frames not needed): child
        visits / sum and root value; also keeps the decision-time fields ``append`` will take (action, mask, to_play, timestep).
        ``sampled_actions`` [n, K, D] / ``improved_policy`` [n, A]: the extra argument the sampled / Gumbel families store
        (taken from the row's extra block by default, shard.unpack_rows)."""
        ids = self._ids(env_ids)
        rows = np.asarray(rows)
        t = self._stats[ids]
        A, H = self.A, shard.HEADER
        self.child_visits[ids, t] = rows[:, H:H + A]
        self.n_legal[ids, t] = rows[:, shard.F_N_LEGAL].astype(np.int64)
        self.root_value[ids, t] = rows[:, shard.F_ROOT_VALUE]
        self.predicted_value[ids, t] = rows[:, shard.F_PRED_VALUE]
        self.entropy[ids, t] = rows[:, shard.F_ENTROPY]
        pos = rows[:, shard.F_ACTION].astype(np.int64)
        if self.sampled_actions is None:
            self.action[ids, t] = pos
        self.action_mask[ids, t] = rows[:, H + A:H + 2 * A]
        self.to_play[ids, t] = rows[:, shard.F_TO_PLAY].astype(np.int64)
        self.timestep[ids, t] = rows[:, shard.F_TIMESTEP].astype(np.int64)
        if self.sampled_actions is not None:
            n_extra = int(np.prod(self.sampled_shape))
            sa = sampled_actions if sampled_actions is not None else rows[:, H + 2 * A:H + 2 * A + n_extra]
            sa = np.asarray(sa, np.float32).reshape((len(ids),) + self.sampled_shape)
            self.sampled_actions[ids, t] = sa
            chosen = sa[np.arange(len(sa)), pos]               # the action itself: [n, D]
            self.action[ids, t] = chosen if self.continuous else chosen[:, 0].astype(np.int64)
        if self.improved is not None:
            ip = improved_policy if improved_policy is not None else rows[:, H + 2 * A:H + 3 * A]
            self.improved[ids, t] = np.asarray(ip, np.float32).reshape(len(ids), A)
        self._stats[ids] = t + 1

    def append(self, next_observations, rewards, env_ids=None):
        """the environment-side half of GameSegment.append (:158-182): o_{t+1} and r_t of the transition whose decision-time
        fields came with ``store_search_stats_rows``; also the collector's observation window (muzero_collector.py:641)"""
        ids = self._ids(env_ids)
        t = self.len[ids]
        frames = np.asarray(next_observations, np.float32).reshape((len(ids),) + self.frame_shape)  # a view when already float32
        self._frames.append((None if env_ids is None else ids.copy(), frames, t.copy(), self._gen[ids].copy()))
        self.reward[ids, t] = np.asarray(rewards, np.float32)
        self.len[ids] = t + 1

    def window(self, env):
        """the collector's observation_window_stack of one env (muzero_collector.py:641): the ``frame_stack_num`` newest observations
        of its current segment, [frame_stack_num, *frame_shape].  Built on demand from the frames kept by reference -- the per-step
        path never shifts or copies a window"""
        n, g, s = int(self.len[env]), self._gen[env], self.stack
        out = np.zeros((s,) + self.frame_shape, np.float32)
        need = min(s, n)
        if n < s:
            out[:s - n] = self._init_obs[env, n:]
        for fi, fr, pos, gen in reversed(self._frames):
            if need == 0:
                break
            k = env if fi is None else (np.nonzero(fi == env)[0][0] if (fi == env).any() else -1)
            if k < 0 or gen[k] != g or int(pos[k]) < n - s:
                continue
            out[s - (n - int(pos[k]))] = fr[k]
            need -= 1
        return out

    def is_full(self):
        """GameSegment.is_full (:370-377), per env"""
        return self.len >= self.L

    # ---- segment hand-over (muzero_collector.py:308-410, 649-692)
    def _priorities(self, env):
        """_compute_priorities (:308-334): L1 distance of predicted and searched root values + 1e-6, or None (= the buffer's maximum)"""
        if not self.use_priority:
            return None
        n = int(self._stats[env])
        return np.abs(self.predicted_value[env, :n] - self.root_value[env, :n]) + np.float32(1e-6)

    def _pad_and_save(self, env, done):
        """pad_and_save_last_trajectory (:336-410): the waiting segment of ``env`` takes the first entries of the current one
        (pad_over, game_segment.py:183-234: ``pad`` observations / actions / child visits / root values, ``pad - 1`` rewards), is
        converted to arrays and pooled"""
        last, cur = self._last[env], self.to_arrays(env)
        p, s = self.pad, self.stack
        last["valid_transition_count"] = min(len(last["action_segment"]), self.L)
        last["obs_segment"] = np.concatenate([last["obs_segment"], cur["obs_segment"][s:s + p]], 0)
        last["reward_segment"] = np.concatenate([last["reward_segment"], cur["reward_segment"][:p - 1]], 0)
        last["action_segment"] = np.concatenate([last["action_segment"], cur["action_segment"][:p]], 0)
        last["root_value_segment"] = np.concatenate([last["root_value_segment"], cur["root_value_segment"][:p]], 0)
        a, b = last["child_visit_segment"], cur["child_visit_segment"][:p]
        if a.dtype == object or b.dtype == object or (a.ndim == 2 and b.ndim == 2 and a.shape[1] != b.shape[1]):
            merged = np.empty(len(a) + len(b), dtype=object)   # variable action spaces: dtype=object like the reference (:316-321)
            for k, row in enumerate(list(a) + list(b)):
                merged[k] = list(row)
            if len(merged) and all(len(x) == len(merged[0]) for x in merged):
                merged = np.array([list(x) for x in merged])
            last["child_visit_segment"] = merged
        else:
            last["child_visit_segment"] = np.concatenate([a, b.reshape((len(b),) + a.shape[1:])], 0) if len(b) else a
        if self.improved is not None:
            last["improved_policy_probs"] = np.concatenate([last["improved_policy_probs"], cur["improved_policy_probs"][:p]], 0)
        self.pool.append((last, self._last_pri[env], bool(done)))
        self._last[env], self._last_pri[env] = None, None

    def rollover(self, done=None, reset_observations=None):
        """The collector's segment logic for every env, after this step's ``append`` (muzero_collector.py:649-692 inside the per-env
        loop): a FULL segment pads and pools its predecessor, takes its place as the waiting segment (with its priorities) and a new
        segment starts from the observation window; a DONE env pads and pools the waiting segment, pools the current one (if it holds
        a transition) and -- when ``reset_observations`` ([n_env | n_done, *frame_shape], the first observation of the next episode)
        is given -- starts a fresh episode.  Returns the number of segments pooled."""
        before = len(self.pool)
        done = np.zeros(self.n_env, bool) if done is None else np.asarray(done, bool)
        flag = np.zeros(self.n_env, bool) if self.ignore_done else done   # what the pool records (muzero_collector.py:621); the episode still ends
        full = self.is_full()
        n_done = 0
        for env in np.nonzero(full | done)[0]:   # env by env, like the collector's loop: the pool keeps its order
            if full[env]:
                if self._last[env] is not None:
                    self._pad_and_save(env, flag[env])
                pri = self._priorities(env)
                self._last[env] = self.to_arrays(env)
                self._last_pri[env] = pri
                self.reset(self._last[env]["obs_segment"][-self.stack:][None], env_ids=[env])   # = the observation window
            if done[env]:
                if self._last[env] is not None:
                    self._pad_and_save(env, flag[env])
                pri = self._priorities(env)
                seg = self.to_arrays(env)
                seg["valid_transition_count"] = min(len(seg["action_segment"]), self.L)
                if len(seg["reward_segment"]) > 0:
                    self.pool.append((seg, pri, bool(flag[env])))
                if reset_observations is not None:
                    ro = np.asarray(reset_observations, np.float32)
                    frame = ro[env] if ro.shape[0] == self.n_env else ro[n_done]
                    self.reset(np.repeat(frame.reshape((1, 1) + self.frame_shape), self.stack, 1), env_ids=[env])
                self._last[env], self._last_pri[env] = None, None
                n_done += 1
        return len(self.pool) - before

    def drain_pool(self):
        """(segments, [{'priorities', 'done', 'unroll_plus_td_steps'}]) like the collector's return_data (:700-710); empties the pool"""
        segs = [p[0] for p in self.pool]
        meta = [dict(priorities=p[1], done=p[2], unroll_plus_td_steps=self.pad) for p in self.pool]
        self.pool = []
        return segs, meta

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
        g = self._gen[env]
        for fi, fr, pos, gen in self._frames[int(self._first[env]) - self._base:]:   # (older entries: previous segments of this env)
            if fi is None:
                if gen[env] == g:
                    obs[self.stack + int(pos[env])] = fr[env]
            else:
                k = np.nonzero(fi == env)[0]
                if k.size and gen[k[0]] == g:
                    obs[self.stack + int(pos[k[0]])] = fr[k[0]]
        out = dict(obs_segment=obs, action_segment=self.action[env, :n].copy(),
                   reward_segment=self.reward[env, :n].copy(), child_visit_segment=child,
                   root_value_segment=self.root_value[env, :ns].copy(), action_mask_segment=self.action_mask[env, :n].copy(),
                   to_play_segment=self.to_play[env, :n].copy(), timestep_segment=self.timestep[env, :n].copy())
        if self.sampled_actions is not None:
            out["root_sampled_actions"] = self.sampled_actions[env, :ns].copy()
        if self.improved is not None:
            out["improved_policy_probs"] = self.improved[env, :ns].copy()
        return out
