"""The collect loop of lzero/worker/muzero_collector.py::MuZeroCollector.collect (:416-760) for a VECTORISED environment: the caller
side of the hot path.  Per env-step the reference walks its ready envs in Python -- unpack the policy's dict, two
``GameSegment`` calls, the observation window, is_full / done hand-over -- here a step is one ``forward_collect_rows`` (search +
``select_action`` + packed env-step rows written by the device), one ``env.step`` over all envs, two array writes
(``GameSegmentBatch.store_search_stats_rows`` / ``append``) and ``rollover`` for the envs whose segment filled up or whose episode
ended.  What it returns is what the reference returns: ``[segments, [{'priorities', 'done', 'unroll_plus_td_steps'}]]``, segments in
the order the reference's per-env loop pools them (``game_segment_to_array`` field set, as dicts).

Environment protocol (``env``; no DI-engine env manager here -- the reference's ``ready_obs`` / ``step`` dictionaries keyed by env id
become arrays over all ``env.env_num`` envs):
    reset() -> obs                       obs = {'observation': [n, *frame_shape] newest frame, 'action_mask': [n, A], 'to_play': [n],
                                                 'timestep': [n] (optional)}
    step(actions [n], active [n] bool) -> (obs, reward [n], done [n], info)
        rows of inactive envs are ignored; for a finished env ``obs`` holds the TERMINAL observation (it is appended to the segment
        like any other, muzero_collector.py:616-620) and ``info['reset_obs']`` -- same keys as ``obs``, rows valid where ``done`` --
        the first observation of its next episode (DI-engine's env manager resets a finished env by itself; :707-727 reads it from
        ``ready_obs``); ``info['eval_episode_return']`` [n] is read where ``done``.

Policy protocol: ``policy.forward_collect_rows(data, action_mask, rows_out, temperature=, to_play=, timestep=, frame_floats=,
epsilon=) -> header [n, 8 + 2 A (+ extra)]`` (lightzero_amd.policy.efficientzero.EfficientZeroPolicy; rows_out may be None for a
host-only policy).  ``data`` is the stacked observation ``[n, frame_stack_num * C, H, W]`` (``prepare_observation`` for conv models,
or ``[n, frame_stack_num * D]`` for vector observations), kept on ``device`` across steps: only the newest frame of every env is
uploaded per step, the stack is shifted where it lives.
Not here: the PPO / pure-policy branches, chance labels, task ids, logging, DDP statistics."""
import numpy as np

from .. import shard
from ..mcts.buffer.game_segment import GameSegmentBatch


def _g(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


class _Group(object):
    """the state of one collect() call over one vectorised env: everything muzero_collector.py:470-512 initialises, as arrays"""

    def __init__(self, col, env, policy, n_episode, policy_kwargs):
        cfg = col._cfg
        self.col, self.env, self.policy, self.n_episode = col, env, policy, int(n_episode)
        n = self.n = int(env.env_num)
        assert n_episode >= n, "Please ensure n_episode (%d) >= env_num (%d)." % (n_episode, n)
        self.temperature, self.epsilon = policy_kwargs.get("temperature", 1.0), policy_kwargs.get("epsilon", 0.0)
        A = col._A
        obs = env.reset()
        frames = np.asarray(obs["observation"], np.float32)
        self.frame_shape = tuple(frames.shape[1:])
        self.F = int(np.prod(self.frame_shape))
        self.sampled = bool(_g(cfg, "sampled_algo", False))
        self.continuous = self.sampled and bool(_g(_g(cfg, "model", {}), "continuous_action_space", False))
        K = self.K = int(_g(_g(cfg, "model", {}), "num_of_sampled_actions", 0) or 0)
        D = self.D = (int(_g(_g(cfg, "model", {}), "action_space_size", 0) or 0) if self.continuous else 1) if self.sampled else 0
        AW = self.AW = K if self.sampled else A            # width of the row's visit-count block
        self.batch = GameSegmentBatch(n, AW, col._L, self.frame_shape, frame_stack_num=col._stack,
                                      num_unroll_steps=int(_g(cfg, "num_unroll_steps")), td_steps=int(_g(cfg, "td_steps")),
                                      sampled_actions_shape=(K, D) if self.sampled else None, improved_policy=bool(_g(cfg, "gumbel_algo", False)),
                                      use_priority=bool(_g(cfg, "use_priority", False)),
                                      use_max_priority_for_new_data=bool(_g(cfg, "use_max_priority_for_new_data", False)),
                                      ignore_done=bool(_g(cfg, "ignore_done", False)), continuous_action_space=self.continuous)
        self.batch.reset(np.repeat(frames[:, None], col._stack, 1))
        self.st = col._stack_init(frames)
        self.mask = np.asarray(obs["action_mask"], np.float32).copy()
        self.to_play = np.asarray(obs["to_play"]).astype(np.int64).copy()
        self.timestep = np.asarray(obs.get("timestep", np.full(n, -1))).astype(np.int64).copy()
        extra = K * D if self.sampled else (A if _g(cfg, "gumbel_algo", False) else 0)   # root_sampled_actions / improved_policy_probs block
        self.rows_out = None
        if col._rows_on_device:
            import torch
            self.rows_out = torch.zeros(n, shard.row_width(AW, self.F, extra), device=col._device)
        self.active = np.ones(n, bool)                  # ready_env_id (:513-516): every env starts one episode ...
        self.remain_episode = self.n_episode - n        # ... and a finished env starts another one while episodes remain
        self.eps_steps, self.entropies = np.zeros(n, np.int64), np.zeros(n, np.float64)
        self.collected_episode = self.collected_step = self.loop_steps = 0
        self.episode_info = []
        self.done = False
        self.rs = None   # this group's host-side random stream (pipelined groups only, MuZeroVectorCollector.collect)

    def run_policy(self):
        """the policy forward of this step (search + select_action + rows on the device); no env or segment state is touched"""
        return self.policy_end(self.policy_begin())

    def policy_begin(self):
        """ENQUEUE this step's policy forward (representation network, noise, the simulations, select_action + rows) and return a ticket
        without waiting for the device (policies without a split forward run it whole in ``policy_end``)"""
        c = self.col
        if c._device is not None and self.rows_out is not None:
            import torch
            torch.cuda.set_device(self.rows_out.device)   # a worker thread starts on device 0: this rank's device is the rows' device
        st = self.st
        if len(self.frame_shape) == 3:   # image frames [C, H, W]: stacked along the channel axis (prepare_observation, 'conv')
            ch, h, w = self.frame_shape
            data = st.reshape(self.n, c._stack * ch, h, w)
        else:
            data = st.reshape(self.n, -1)
        kw = dict(temperature=self.temperature, to_play=self.to_play.tolist(), timestep=self.timestep.astype(np.int32), frame_floats=self.F, epsilon=self.epsilon)
        split = hasattr(self.policy, "forward_collect_rows_begin")

        def begin():
            if split:
                return ("ticket", self.policy.forward_collect_rows_begin(data, self.mask, self.rows_out, **kw))
            return ("call", (data, kw))
        if self.rs is None:      # a single group: the global np.random stream, like the reference
            return begin()
        from .. import _lib as L
        with L.random_source(self.rs):   # several groups: this group's own host-side stream
            return begin()

    def policy_end(self, ticket):
        """wait for the forward ``policy_begin`` enqueued -> the [n, 8 + 2 A (+ extra)] header block"""
        kind, t = ticket

        def end():
            if kind == "ticket":
                return np.asarray(self.policy.forward_collect_rows_end(t))
            data, kw = t
            return np.asarray(self.policy.forward_collect_rows(data, self.mask, self.rows_out, **kw))
        if self.rs is None:
            return end()
        from .. import _lib as L
        with L.random_source(self.rs):
            return end()

    def step_envs(self, header):
        """First half of a collector step: env.step with the chosen actions and everything that determines the NEXT policy forward -- the
        masks / to_play / timesteps the policy will see, the shifted observation stack (reset observations for finished envs), which envs
        stay active, whether the call is complete (``self.done``).  Returns the record ``book`` needs.  Splitting the step here lets the
        collector launch the next forward (3.3 ms on the device) BEFORE the segment bookkeeping of this step (array copies of 9.4 MB of
        frames, rollovers: ~0.5 ms of host time) instead of after it."""
        n, AW, K, D, active = self.n, self.AW, self.K, self.D, self.active
        mask, to_play, timestep = self.mask, self.to_play, self.timestep
        actions = header[:, shard.F_ACTION].astype(np.int64)
        if self.sampled:   # word 0 is the position among the K sampled actions; the action is that entry of the extra block
            sa = header[:, shard.HEADER + 2 * AW:shard.HEADER + 2 * AW + K * D].reshape(n, K, D)[np.arange(n), actions]
            actions = sa if self.continuous else sa[:, 0].astype(np.int64)
        was_active = active.copy()
        obs, reward, done, info = self.env.step(actions, was_active)
        done = np.asarray(done, bool) & was_active
        ids = None if was_active.all() else np.nonzero(was_active)[0]
        sel = slice(None) if ids is None else ids
        nxt = np.asarray(obs["observation"], np.float32)
        mask[sel] = np.asarray(obs["action_mask"], np.float32)[sel]
        to_play[sel] = np.asarray(obs["to_play"]).astype(np.int64)[sel]
        if "timestep" in obs:
            timestep[sel] = np.asarray(obs["timestep"]).astype(np.int64)[sel]
        self.collected_step += int(was_active.sum())
        self.loop_steps += 1
        fin = np.nonzero(done)[0]
        reset_frames, ro = None, None
        if fin.size:
            ro = info["reset_obs"]
            reset_frames = np.asarray(ro["observation"], np.float32)
        self.st = self.col._stack_push(self.st, nxt, fin, reset_frames[fin] if fin.size else None)
        for e in fin:
            self.collected_episode += 1
            mask[e] = np.asarray(ro["action_mask"], np.float32)[e]
            to_play[e] = int(np.asarray(ro["to_play"])[e])
            timestep[e] = int(np.asarray(ro["timestep"])[e]) if "timestep" in ro else -1
            active[e] = False
        for e in fin:   # (:513-516 of the next iteration) a finished env takes one of the remaining episodes, lowest id first
            if self.remain_episode > 0:
                active[e] = True
                self.remain_episode -= 1
        self.done = self.collected_episode >= self.n_episode
        return dict(header=header, ids=ids, sel=sel, nxt=nxt, reward=np.asarray(reward, np.float32), done=done, fin=fin, reset_frames=reset_frames,
                    returns=np.asarray(info["eval_episode_return"]) if fin.size else None)

    def book(self, rec):
        """Second half: the bookkeeping of muzero_collector.py:588-735 for the step ``step_envs`` just made -- search statistics and the new
        observation into the segments, segment hand-over / episode ends in the reference's order, episode statistics.  Touches nothing
        the policy forward reads, so it may run while the next forward is on the device."""
        header, ids, sel, batch = rec["header"], rec["ids"], rec["sel"], self.batch
        # the decision-time fields of the rows are the mask / to_play / timestep the policy saw (muzero_collector.py:616-620)
        batch.store_search_stats_rows(header[sel], env_ids=ids)
        batch.append(rec["nxt"][sel], rec["reward"][sel], env_ids=ids)
        self.eps_steps[sel] += 1
        self.entropies[sel] += header[sel, shard.F_ENTROPY]
        # ---- segment hand-over and episode ends, env by env in the reference's order (:649-735)
        batch.rollover(rec["done"], reset_observations=rec["reset_frames"])
        for e in rec["fin"]:
            self.episode_info.append(dict(reward=float(rec["returns"][e]), step=int(self.eps_steps[e]),
                                          visit_entropy=float(self.entropies[e] / self.eps_steps[e]) if self.eps_steps[e] else 0.0))
            self.eps_steps[e], self.entropies[e] = 0, 0.0

    def finish(self, header):
        """env.step + the bookkeeping of muzero_collector.py:588-735 for this step; returns True when n_episode episodes are in"""
        self.book(self.step_envs(header))
        return self.done


class MuZeroVectorCollector(object):
    def __init__(self, env, policy, policy_config, device=None, rows_on_device=True, pipeline=True):
        """``env`` / ``policy``: one vectorised env and one policy -- or two lists of the same length (env GROUPS, one policy object per
        group, all on the same engine model): ``collect`` then pipelines the groups -- every group keeps one forward ENQUEUED on the
        engine's stream (``forward_collect_rows_begin`` does not wait for the device), so while the device searches for one group the
        host steps the environments of the other and does its segment bookkeeping, and the GPU goes from one group's search straight
        into the next one's.  Every group by itself runs
        the loop of the single-group form (same transcript -> same pooled segments; with more than one group every group draws its host-side
        random numbers from a stream of its own, seeded from np.random at the start of ``collect``, so a seeded run is reproducible as long as the
        envs do not share the global np.random stream across groups either); everything runs on the calling thread and the engine's one
        stream serialises the groups' forwards in the order they were enqueued."""
        self._groups_env = list(env) if isinstance(env, (list, tuple)) else [env]
        self._groups_policy = list(policy) if isinstance(policy, (list, tuple)) else [policy]
        assert len(self._groups_env) == len(self._groups_policy), "one policy object per env group"
        self._cfg = policy_config
        m = _g(policy_config, "model", {})
        self._A = int(_g(m, "action_space_size"))
        self._stack = int(_g(m, "frame_stack_num", 1))
        self._L = int(_g(policy_config, "game_segment_length"))
        self.unroll_plus_td_steps = int(_g(policy_config, "num_unroll_steps")) + int(_g(policy_config, "td_steps"))
        self._default_n_episode = _g(policy_config, "n_episode", None)
        self._device = device
        self._rows_on_device = rows_on_device and device is not None
        self._pipeline = bool(pipeline)   # single group: enqueue the next forward before this step's segment bookkeeping (see collect)
        self.episode_info = []          # {'reward', 'step', 'visit_entropy'} per finished episode (muzero_collector.py:659-666)
        self.total_envstep_count = 0
        self.total_episode_count = 0
        self.total_loop_steps = 0       # policy forwards issued (every one over all envs of its group, also while some wait for the last episodes)
        self.group_results = []        # per group of the last collect(): (segments, meta, episode_info)

    # ---- stacked observation on the device (or on the host for a host-only policy)
    def _stack_init(self, frames):
        st = np.repeat(frames[:, None], self._stack, 1)   # [n, stack, *frame_shape]: the first frame repeated (:479-482)
        if self._device is None:
            return st
        import torch
        return torch.from_numpy(np.ascontiguousarray(st)).to(self._device)

    def _stack_push(self, st, frames, reset_rows=None, reset_frames=None):
        if self._device is None:
            st = np.concatenate([st[:, 1:], frames[:, None]], 1)
            if reset_rows is not None and len(reset_rows):
                st[reset_rows] = np.repeat(reset_frames[:, None], self._stack, 1)
            return st
        import torch
        new = torch.from_numpy(np.ascontiguousarray(frames)).to(self._device, non_blocking=True)
        st = torch.cat([st[:, 1:], new[:, None]], 1)
        if reset_rows is not None and len(reset_rows):
            rf = torch.from_numpy(np.ascontiguousarray(reset_frames)).to(self._device)
            st[torch.as_tensor(reset_rows, device=self._device)] = rf[:, None].expand(-1, self._stack, *rf.shape[1:])
        return st

    @staticmethod
    def _drain_ticket(group, ticket):
        """wait for a forward that will not be consumed (error path) so that its roots handle is reusable; never raises"""
        try:
            group.policy_end(ticket)
        except Exception:
            pass

    def collect(self, n_episode=None, train_iter=0, policy_kwargs=None):
        """``n_episode``: episodes to collect in total; with env groups it is split evenly (the remainder to the first groups) and every
        group needs at least ``env_num`` of them, like the reference's single env manager (:449)."""
        if n_episode is None:
            if self._default_n_episode is None:
                raise RuntimeError("Please specify `n_episode` for collection.")
            n_episode = self._default_n_episode
        G = len(self._groups_env)
        share = [n_episode // G + (1 if g < n_episode % G else 0) for g in range(G)]
        groups = [_Group(self, e, p, k, policy_kwargs or {}) for e, p, k in zip(self._groups_env, self._groups_policy, share)]
        if G > 1:
            # Pipelined groups: a group's policy forward (Dirichlet noise, eps-greedy draws, the seeds of the device-side select_action)
            # runs on a worker thread while the main thread steps the other group's envs -- which typically draw from np.random too.
            # Two threads interleaving on the global stream would make a seeded run irreproducible, so every group gets its own
            # stream, seeded HERE on the main thread from np.random (np.random.seed governs it).  A single group keeps the global stream.
            for g in groups:
                g.rs = np.random.RandomState(int(np.random.randint(0, 2 ** 31 - 1)))
        if G == 1 and not self._pipeline:
            g = groups[0]
            while not g.finish(g.run_policy()):
                pass
        elif G == 1:
            # One group, software-pipelined on ONE thread: the forward of step t + 1 is enqueued (policy_begin returns without waiting
            # for the device) as soon as step t's env.step has produced its inputs, step t's segment bookkeeping runs while the
            # device searches, then policy_end waits for the rows.  Same calls on the same data in the same order per object; the
            # host-side random draws happen in the order of the plain loop (noise and seeds in begin, eps-greedy in end, the
            # environments afterwards).
            g = groups[0]
            tk = g.policy_begin()
            try:
                while True:
                    header = g.policy_end(tk)
                    tk = None
                    rec = g.step_envs(header)
                    if not g.done:
                        tk = g.policy_begin()
                    g.book(rec)
                    if g.done:
                        break
            finally:
                # an exception in env.step / the bookkeeping must not leave a forward in flight: its roots handle would stay "rows pending"
                # in the policy's handle cache and every later collect() would be refused
                if tk is not None:
                    self._drain_ticket(g, tk)
        else:
            # Several env groups: every live group has ONE forward enqueued on the engine's stream at any time, so the device goes from
            # one group's search straight into the next one's while the host -- one thread -- waits for the oldest forward (its own
            # event: lz_roots_collect_rows_end), steps that group's environments, enqueues its next forward behind the others' and does
            # its segment bookkeeping.  (Round 4 kept one forward in flight on a worker thread: the device idled from one group's
            # read-back to the next group's launch, ~0.4 ms of every 3.3 ms step.)
            import collections
            queue = collections.deque()
            try:
                for g in groups:
                    queue.append((g, g.policy_begin()))
                while queue:
                    g, tk = queue.popleft()
                    rec = g.step_envs(g.policy_end(tk))
                    if not g.done:
                        queue.append((g, g.policy_begin()))
                    g.book(rec)
            finally:
                while queue:   # (only after an exception: every other group's forward in flight is waited for and dropped)
                    self._drain_ticket(*queue.popleft())
        self.group_results = []
        segs_all, meta_all = [], []
        for g in groups:
            segs, meta = g.batch.drain_pool()
            self.group_results.append((segs, meta, g.episode_info))
            segs_all += segs
            meta_all += meta
            self.episode_info += g.episode_info
            self.total_envstep_count += g.collected_step
            self.total_episode_count += g.collected_episode
            self.total_loop_steps += g.loop_steps
        return [segs_all, meta_all]
