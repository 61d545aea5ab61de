"""Host-side mirror of the reference's Cython tree modules on top of the C ABI.

Same names, argument meaning and return values as lzero/mcts/ctree/ctree_efficientzero/ez_tree.pyx
and lzero/mcts/ctree/ctree_muzero/mz_tree.pyx; the trees themselves live in HBM and every call
runs HIP kernels (lightzero_amd/csrc/lz_tree.hip).  Differences, all forced by the flat node pool:

* ``Roots(root_num, legal_actions_list, action_space_size=None, max_simulations=None)`` -- the node
  pool is sized at construction; when the two extra arguments are omitted the action-space size is
  taken at ``prepare`` time from the width of ``policy_logits_pool`` and ``max_simulations``
  defaults to ``DEFAULT_MAX_SIMULATIONS``.
* The min-max statistics live with the roots; a ``MinMaxStatsList`` is bound to the ``Roots`` it is
  first used with (the reference constructs a fresh list for every search, mcts_ctree.py:778-779).
* Tie-breaking defaults to the reference's stochastic rule (uniform over the tie list) for the
  EfficientZero tree and follows the ``deterministic`` flag for the MuZero tree; call
  ``Roots.set_tiebreak(0)`` for the deterministic first-arg-max rule.
"""
import collections
import ctypes
import os

import numpy as np

from ... import _lib as L

DEFAULT_MAX_SIMULATIONS = 512

# The reference builds a fresh ``Roots`` for every forward (efficientzero.py:605).  A device handle owns HBM pools (trees, and
# after an inference the latent / LSTM pools), so handles released by a dead ``Roots`` object are parked here -- keyed by
# (engine, variant, root_num, action space, max_simulations) -- and re-armed (lz_roots_reset) by the next ``Roots`` of that shape
# instead of being freed and re-allocated every env-step.
# Bounded: at most _HANDLE_CACHE_DEPTH handles per shape and _HANDLE_CACHE_MAX handles in total (LZ_HANDLE_CACHE_MAX, default 6);
# beyond that the least recently parked handle is destroyed (its trees and latent / LSTM pools go back to the device).  A driver that
# builds a fresh Roots per forward with a varying ready-env count therefore keeps a few pool sets, not one or two per batch size
# it has ever seen (120 MB each at 256 roots x 50 simulations).  ``flush_handle_cache()`` frees everything that is parked.
_HANDLE_CACHE = {}
_HANDLE_CACHE_DEPTH = 2
_HANDLE_CACHE_MAX = max(0, int(os.environ.get("LZ_HANDLE_CACHE_MAX", "6")))
_PARK_ORDER = collections.OrderedDict()   # handle address -> key, oldest first


def _engine_key(engine):
    return getattr(engine, "value", engine)


def _haddr(h):
    return getattr(h, "value", h)


def _unpark(key):
    """the most recently parked handle of this shape, or None"""
    parked = _HANDLE_CACHE.get(key)
    if not parked:
        return None
    h, seed = parked.pop()
    _PARK_ORDER.pop(_haddr(h), None)
    if not parked:
        _HANDLE_CACHE.pop(key, None)
    return h, seed


def _park(key, h, seed):
    """park a released handle; returns False when the caller has to destroy it (cache disabled / this shape is full)"""
    if _HANDLE_CACHE_MAX == 0 or len(_HANDLE_CACHE.get(key, ())) >= _HANDLE_CACHE_DEPTH:
        return False
    _HANDLE_CACHE.setdefault(key, []).append((h, seed))
    _PARK_ORDER[_haddr(h)] = key
    while len(_PARK_ORDER) > _HANDLE_CACHE_MAX:   # evict the least recently parked handle, whatever its shape
        addr, k = _PARK_ORDER.popitem(last=False)
        lst = _HANDLE_CACHE.get(k, [])
        for i, (hh, _) in enumerate(lst):
            if _haddr(hh) == addr:
                lst.pop(i)
                L.lib().lz_roots_destroy(hh)
                break
        if not lst:
            _HANDLE_CACHE.pop(k, None)
    return True


def flush_handle_cache(engine=None):
    """destroy every parked roots handle (of one engine, or of all): their HBM goes back to the device.  Returns the count."""
    ev = None if engine is None else _engine_key(engine)
    n = 0
    for key in [k for k in _HANDLE_CACHE if ev is None or (k and k[0] == ev)]:
        for h, _ in _HANDLE_CACHE.pop(key):
            _PARK_ORDER.pop(_haddr(h), None)
            L.lib().lz_roots_destroy(h)
            n += 1
    return n


def handle_cache_size():
    return len(_PARK_ORDER)


def _purge_engine(engine_value):
    """an engine is about to be destroyed (lightzero_amd._lib.OwnedEngine): its parked roots handles go first"""
    flush_handle_cache(engine_value)


L._engine_death_hooks.append(_purge_engine)


def collect_rows_ex(roots, A_row, temperature, deterministic, d_rows_ptr, row_words, frame_floats, discount=0.997, timestep=None, seed=None,
                    policy_width=None, d_obs_ptr=None):
    """lz_roots_collect_rows_ex for any roots handle: (header [B, 8 + 2 A_row + extra] on the host, root policy logits); the extra
    block holds root_sampled_actions (sampled roots) / improved_policy_probs (Gumbel roots), see include/lz_mi355.h"""
    B = roots.root_num
    E = int(L.lib().lz_rows_extra_words(roots._h))
    hdr = np.zeros((B, 8 + 2 * A_row + E), np.float32)
    lg = np.zeros((B, policy_width or A_row), np.float32)
    if seed is None:
        seed = int(L.rs().randint(0, 2 ** 62))
    ts = None if timestep is None else L.i32(timestep)
    L.check(L.lib().lz_roots_collect_rows_ex(roots._h, float(temperature), 1 if deterministic else 0, int(seed), float(discount), d_obs_ptr,
                                             int(frame_floats), None if ts is None else ts.ctypes.data, d_rows_ptr, int(row_words), hdr,
                                             lg.ctypes.data))
    return hdr, lg


def make_module(variant, has_deterministic_flag):
    class MinMaxStatsList(object):
        def __init__(self, num):
            self.num = int(num)
            self._delta = 0.0
            self._bound = None

        def set_delta(self, value_delta_max):
            self._delta = float(value_delta_max)

    class ResultsWrapper(object):
        def __init__(self, num):
            self.num = int(num)
            self._search_lens = []
            self._roots = None

        def get_search_len(self):
            return self._search_lens

    class Roots(object):
        def __init__(self, root_num, legal_actions_list, action_space_size=None, max_simulations=None, engine=None):
            self.root_num = int(root_num)
            if len(legal_actions_list) != self.root_num:
                raise ValueError("legal_actions_list must have root_num entries")
            self._legal = [[int(a) for a in l] for l in legal_actions_list]
            self._A = action_space_size
            self._S = int(max_simulations) if max_simulations else DEFAULT_MAX_SIMULATIONS
            self._engine = engine
            self._h = None
            self._tiebreak = None  # None -> module default
            # the reference seeds its tie-break stream from the clock (srand(tv_usec) per traverse, cnode.cpp:901); here it follows
            # np.random's state and the rank (set_tiebreak(mode, seed) / the policies' ``mcts_seed`` pin it)
            self._seed = None      # drawn (L.process_seed) when the device handle is created, or inherited from a re-armed handle
            self._inferred_by = None
            self._prep_args = None  # the last host-side prepare (replayed when the handle has to be re-created, _adopt)
            self._touched = False  # a prepare / inference has used the device handle

        @property
        def num(self):
            return self.root_num

        def _bind_engine(self, engine):
            """an engine model is about to run on these roots: they must live on the model's engine"""
            mine = self._engine if self._engine is not None else L.default_engine()
            if self._h is not None and getattr(mine, "value", mine) != getattr(engine, "value", engine):
                if self._touched:
                    raise L.LzError("these roots already hold a search on another engine than the model's: build them with "
                                    "Roots(..., engine=model.engine) / MCTSCtree.roots(..., engine=model.engine)")
                self.clear()  # an empty node pool: re-create it where the model lives
            self._engine = engine
            self._touched = True

        def _ensure(self, A):
            if self._h is not None:
                return
            if self._A is None:
                self._A = int(A)
            if self._A != int(A):
                raise ValueError("policy_logits width %d != action_space_size %d" % (A, self._A))
            eng = self._engine if self._engine is not None else L.default_engine()
            if self._legal is None:  # the lists were last given as a mask (reset_mask): its flattened form is kept
                cnt, flat = self._legal_cnt, self._legal_flat
            else:
                cnt = L.i32([len(l) for l in self._legal])
                flat = L.i32([a for l in self._legal for a in l] or [0])
            self._key = (_engine_key(eng), variant, self.root_num, self._A, self._S)
            parked = _unpark(self._key)
            if parked:
                # an unpinned seed stays the handle's: the captured search graph (keyed by it) is replayed instead of re-captured,
                # and the random streams still advance through the device-resident epoch that every prepare bumps
                h, seed0 = parked
                if self._seed is None:
                    self._seed = seed0
                else:   # a pinned seed means the same streams whatever the handle did before
                    L.check(L.lib().lz_roots_reseed(h, self._seed))
                L.check(L.lib().lz_roots_reset(h, flat, cnt))
            else:
                h = L.P()
                L.check(L.lib().lz_roots_create(eng, variant, self.root_num, self._A, self._S, flat, cnt, ctypes.byref(h)))
                if self._seed is None:
                    self._seed = L.process_seed()
            self._h = h
            mode = self._tiebreak if self._tiebreak is not None else (0 if has_deterministic_flag else 1)
            L.check(L.lib().lz_roots_set_tiebreak(self._h, mode, self._seed))

        def reset(self, legal_actions_list, keep_inference=False):
            """Re-arm these roots for a new env-step (what building a fresh ``Roots`` does in the reference) while
            keeping the HBM node / latent pools: same root_num and action space, new legal-action lists.
            ``keep_inference=True``: the engine model's initial_inference for THIS env-step was already launched on these
            roots (the host builds the legal lists while the representation network runs)."""
            if len(legal_actions_list) != self.root_num:
                raise ValueError("legal_actions_list must have root_num entries")
            if keep_inference and self._h is None:
                raise L.LzError("reset(keep_inference=True) needs an engine model's initial_inference on these roots first")
            self._legal = [[int(a) for a in l] for l in legal_actions_list]
            if not keep_inference:
                self._inferred_by = None
            if self._h is not None:
                cnt = L.i32([len(l) for l in self._legal])
                flat = L.i32([a for l in self._legal for a in l] or [0])
                fn = L.lib().lz_roots_reset_keep_inference if keep_inference else L.lib().lz_roots_reset
                L.check(fn(self._h, flat, cnt))
            return self

        def reset_mask(self, action_mask, keep_inference=False):
            """``reset`` from the [root_num, A] action mask itself (the collector's array, muzero_collector.py:533): the legal lists
            are never materialised as Python lists -- one np.nonzero for the whole batch instead of one per env
            (efficientzero.py:595)."""
            m = np.asarray(action_mask) != 0
            if m.shape != (self.root_num, self._A) or self._h is None:
                raise ValueError("reset_mask needs an existing device handle and a [root_num, action_space_size] mask")
            cnt = m.sum(1).astype(np.int32)
            if (cnt == 0).any():
                raise ValueError("every root needs at least one legal action")
            flat = np.ascontiguousarray(np.nonzero(m)[1].astype(np.int32))  # row-major: ascending actions per root
            self._legal = None
            self._legal_cnt, self._legal_flat = cnt, flat   # what _ensure needs to re-create the device handle
            self._n_noise = int(cnt.sum())
            if not keep_inference:
                self._inferred_by = None
            fn = L.lib().lz_roots_reset_keep_inference if keep_inference else L.lib().lz_roots_reset
            L.check(fn(self._h, flat, cnt))
            return self

        def _want_noise(self):
            if self._legal is None:
                return self._n_noise
            return sum(len(l) if l else self._A for l in self._legal)

        def set_tiebreak(self, mode, seed=None):
            """0: first arg-max (deterministic); 1: uniform over the reference's tie list."""
            self._tiebreak = int(mode)
            if seed is not None:
                self._seed = int(seed)
            if self._h is not None:
                L.check(L.lib().lz_roots_set_tiebreak(self._h, self._tiebreak, self._seed))

        def prepare(self, root_noise_weight, noises, value_prefix_pool, policy_logits_pool, to_play_batch):
            logits = L.f32(policy_logits_pool)
            if logits.ndim != 2 or logits.shape[0] != self.root_num:
                raise ValueError("policy_logits_pool must be [root_num][action_space_size]")
            self._ensure(logits.shape[1])
            self._touched = True
            self._prep_args = ("noise", (root_noise_weight, noises, value_prefix_pool, policy_logits_pool, to_play_batch))
            nz = L.f32([x for row in noises for x in row] or [0.0])
            want = self._want_noise()
            if nz.size < want:
                raise ValueError("noises must hold one value per legal action")
            L.check(L.lib().lz_roots_prepare(self._h, float(root_noise_weight), nz.ctypes.data,
                                             L.f32(value_prefix_pool), logits, L.i32(to_play_batch)))

        def prepare_no_noise(self, value_prefix_pool, policy_logits_pool, to_play_batch):
            logits = L.f32(policy_logits_pool)
            if logits.ndim != 2 or logits.shape[0] != self.root_num:
                raise ValueError("policy_logits_pool must be [root_num][action_space_size]")
            self._ensure(logits.shape[1])
            self._touched = True
            self._prep_args = ("no_noise", (value_prefix_pool, policy_logits_pool, to_play_batch))
            L.check(L.lib().lz_roots_prepare(self._h, 0.0, None, L.f32(value_prefix_pool), logits,
                                             L.i32(to_play_batch)))

        def prepare_from_inference(self, root_noise_weight, noises, to_play_batch):
            """Roots.prepare with the policy logits an engine model's initial_inference left in HBM
            (value prefix 0, efficientzero_model.py:238); noises: one list per root over its legal actions."""
            if self._h is None:
                raise L.LzError("prepare_from_inference before an engine model's initial_inference on these roots")
            if isinstance(noises, np.ndarray):  # [root_num][n_legal] rows of equal length
                nz = np.ascontiguousarray(noises, np.float32).reshape(-1)
            else:
                nz = np.concatenate([np.asarray(row, np.float32).reshape(-1) for row in noises]) if len(noises) else np.zeros(1, np.float32)
            want = self._want_noise()
            if nz.size < want:  # the library copies sum(n_legal) floats from this pointer
                raise ValueError("noises must hold one value per legal action (%d < %d)" % (nz.size, want))
            if len(to_play_batch) != self.root_num:
                raise ValueError("to_play_batch must have root_num entries")
            L.check(L.lib().lz_roots_prepare_from_inference(self._h, float(root_noise_weight), nz.ctypes.data,
                                                            L.i32(to_play_batch)))

        def prepare_from_inference_dirichlet(self, root_noise_weight, root_dirichlet_alpha, to_play_batch):
            """Roots.prepare with the logits of the engine model's initial_inference and Dirichlet(alpha) exploration noise drawn
            ON THE DEVICE for every root's legal actions (efficientzero.py:599-605 draws it per env with numpy): no noise array
            is built or uploaded.  Same distribution as the reference's, its own random stream (seed: set_tiebreak / mcts_seed)."""
            if self._h is None:
                raise L.LzError("prepare_from_inference_dirichlet before an engine model's initial_inference on these roots")
            if len(to_play_batch) != self.root_num:
                raise ValueError("to_play_batch must have root_num entries")
            L.check(L.lib().lz_roots_prepare_from_inference_dirichlet(self._h, float(root_noise_weight), float(root_dirichlet_alpha),
                                                                      L.i32(to_play_batch)))

        def prepare_from_inference_no_noise(self, to_play_batch):
            if self._h is None:
                raise L.LzError("prepare_from_inference_no_noise before an engine model's initial_inference on these roots")
            if len(to_play_batch) != self.root_num:
                raise ValueError("to_play_batch must have root_num entries")
            L.check(L.lib().lz_roots_prepare_from_inference(self._h, 0.0, None, L.i32(to_play_batch)))

        def get_distributions(self):
            if self._h is None:
                return [[] for _ in range(self.root_num)]
            out = np.zeros((self.root_num, self._A), np.int32)
            cnt = np.zeros(self.root_num, np.int32)
            L.check(L.lib().lz_roots_get_distributions(self._h, out, cnt))
            return [out[i, :cnt[i]].tolist() for i in range(self.root_num)]

        def get_values(self):
            if self._h is None:
                return [0.0] * self.root_num
            out = np.zeros(self.root_num, np.float32)
            L.check(L.lib().lz_roots_get_values(self._h, out))
            return out.tolist()

        def get_trajectories(self):
            if self._h is None:
                return [[] for _ in range(self.root_num)]
            stride = self._S + 2
            out = np.zeros((self.root_num, stride), np.int32)
            L.check(L.lib().lz_roots_get_trajectories(self._h, out, stride))
            res = []
            for i in range(self.root_num):
                row = out[i].tolist()
                res.append(row[:row.index(-1)])
            return res

        def get_search_results(self, policy_width=None, select=None):
            """After a fused search: (visit counts [B][A] int32, -1 padded; legal counts [B]; root values [B]; predicted root values
            [B]; root policy logits [B][policy_width]) in ONE read-back (lz_roots_get_search_results).
            ``select=(temperature, deterministic[, seed])`` also runs select_action for every root on the device before the
            read-back and appends (action positions [B], entropies [B]) to the returned tuple -- still one synchronisation."""
            B, A = self.root_num, self._A
            dist = np.zeros((B, A), np.int32); cnt = np.zeros(B, np.int32); val = np.zeros(B, np.float32)
            pred = np.zeros(B, np.float32); lg = np.zeros((B, policy_width or A), np.float32)
            if select is None:
                L.check(L.lib().lz_roots_get_search_results(self._h, dist, cnt, val, pred.ctypes.data, lg.ctypes.data))
                return dist, cnt, val, pred, lg
            temperature, deterministic = select[0], select[1]
            seed = select[2] if len(select) > 2 and select[2] is not None else int(L.rs().randint(0, 2 ** 62))
            pos = np.zeros(B, np.int32); ent = np.zeros(B, np.float64)
            L.check(L.lib().lz_roots_get_search_results_select(self._h, dist, cnt, val, pred.ctypes.data, lg.ctypes.data,
                                                               float(temperature), 1 if deterministic else 0, int(seed), pos, ent.ctypes.data))
            return dist, cnt, val, pred, lg, pos, ent

        def collect_rows(self, temperature, deterministic, d_rows_ptr, row_words, frame_floats, timestep=None, seed=None,
                         policy_width=None, d_obs_ptr=None):
            """After a fused search: select_action + the packed env-step rows (lightzero_amd/shard.py schema) written on the device
            into ``d_rows_ptr`` ([root_num][row_words] float32 in HBM); returns (row headers [B, 8 + 2A] on the host, root policy
            logits [B, policy_width]) -- one synchronisation (lz_roots_collect_rows)."""
            B, A = self.root_num, self._A
            hdr = np.zeros((B, 8 + 2 * A), np.float32)
            lg = np.zeros((B, policy_width or A), np.float32)
            if seed is None:
                seed = int(L.rs().randint(0, 2 ** 62))
            ts = None if timestep is None else L.i32(timestep)
            L.check(L.lib().lz_roots_collect_rows(self._h, float(temperature), 1 if deterministic else 0, int(seed), d_obs_ptr,
                                                  int(frame_floats), None if ts is None else ts.ctypes.data, d_rows_ptr, int(row_words),
                                                  hdr, lg.ctypes.data))
            return hdr, lg

        def collect_rows_begin(self, temperature, deterministic, d_rows_ptr, row_words, frame_floats, timestep=None, seed=None,
                               d_obs_ptr=None, discount=0.997, want_logits=False):
            """first half of ``collect_rows`` (lz_roots_collect_rows_begin): select_action + row packing ENQUEUED behind the search, nothing
            waited for -- the caller may enqueue other work (another env group's search) or do host work before ``collect_rows_end``"""
            if seed is None:
                seed = int(L.rs().randint(0, 2 ** 62))
            ts = None if timestep is None else L.i32(timestep)
            L.check(L.lib().lz_roots_collect_rows_begin(self._h, float(temperature), 1 if deterministic else 0, int(seed), float(discount), d_obs_ptr,
                                                        int(frame_floats), None if ts is None else ts.ctypes.data, d_rows_ptr, int(row_words),
                                                        1 if want_logits else 0))

        def collect_rows_end(self, policy_width=None, want_logits=False):
            """second half: waits for the rows enqueued by ``collect_rows_begin`` (their event only) -> (headers [B, 8 + 2A + extra], logits | None)"""
            B, A = self.root_num, self._A
            E = int(L.lib().lz_rows_extra_words(self._h))
            hdr = np.zeros((B, 8 + 2 * A + E), np.float32)
            lg = np.zeros((B, policy_width or A), np.float32) if want_logits else None
            L.check(L.lib().lz_roots_collect_rows_end(self._h, hdr, None if lg is None else lg.ctypes.data))
            return hdr, lg

        def select_action(self, temperature=1, deterministic=True, seed=None):
            """select_action (lzero/policy/utils.py:637-661) for every root on the device: returns (action positions
            [root_num], entropies in bits [root_num]).  deterministic=False draws from N^(1/T) with the engine's
            counter-based generator (``seed``; a fresh one per call when None) instead of np.random."""
            pos = np.zeros(self.root_num, np.int32)
            ent = np.zeros(self.root_num, np.float64)
            if seed is None:
                seed = int(L.rs().randint(0, 2 ** 62))
            L.check(L.lib().lz_roots_select_action(self._h, float(temperature), 1 if deterministic else 0, int(seed), pos, ent))
            return pos, ent

        def get_root_priors(self):
            """priors of the roots' edges after prepare ([root_num, A] by action, 0 where illegal) -- observability"""
            out = np.zeros((self.root_num, self._A), np.float32)
            L.check(L.lib().lz_roots_get_root_priors(self._h, out.reshape(-1)))
            return out

        def get_minmax(self):
            out = np.zeros((self.root_num, 2), np.float32)
            L.check(L.lib().lz_roots_get_minmax(self._h, out))
            return out

        def clear(self, park=False):
            if self._h is not None:
                kept = False
                if park and getattr(self, "_key", None):
                    try:
                        L.check(L.lib().lz_roots_enable_trace(self._h, 0))
                        kept = _park(self._key, self._h, self._seed)
                    except Exception:
                        kept = False
                if not kept:
                    L.lib().lz_roots_destroy(self._h)
                self._h = None

        def __del__(self):
            try:
                self.clear(park=True)   # the next Roots of this shape re-arms the handle (see _HANDLE_CACHE)
            except Exception:
                pass

        # ---- the reference's call order: model.initial_inference(obs) BEFORE the roots exist (efficientzero.py:582-610)
        def _replay_prepare(self):
            kind, args = self._prep_args
            (self.prepare if kind == "noise" else self.prepare_no_noise)(*args)

        def _adopt(self, src_roots, model, num_simulations=None):
            """take over the inference an engine model left in ``src_roots`` (its own handle): these roots were built and prepared
            from host lists afterwards, reference-style.  Moves them to the model's engine / a right-sized node pool first when
            that is needed and possible (the prepare arguments are replayed)."""
            if self._h is None or getattr(self, "_prep_args", None) is None:
                raise L.LzError("search with HBM tokens needs roots.prepare(...) / prepare_no_noise(...) on these roots first")
            mine = self._engine if self._engine is not None else L.default_engine()
            move = _engine_key(mine) != _engine_key(model.engine)
            shrink = num_simulations is not None and self._S >= 2 * int(num_simulations) and self._S == DEFAULT_MAX_SIMULATIONS
            if move or shrink:
                self.clear(park=True)
                self._engine = model.engine
                if shrink:
                    self._S = int(num_simulations)
                self._replay_prepare()
            if src_roots._h is None:   # e.g. evicted from the model's bounded handle cache (EfficientZeroModel._own_roots) since the inference
                raise L.LzError("the HBM token is stale: the roots handle that held this inference was released%s -- call model.initial_inference(obs) "
                                "again (the model keeps the %s most recent (kind, batch size) handles)"
                                % (" by %s's handle cache" % src_roots._evicted_from if getattr(src_roots, "_evicted_from", None) else "",
                                   getattr(model, "_OWN_ROOTS_MAX", "few")))
            L.check(L.lib().lz_roots_adopt_inference(self._h, src_roots._h))
            self._inferred_by = model
            self._touched = True

    def _bind(roots, mm):
        if mm._bound is not roots:
            L.check(L.lib().lz_roots_minmax_reset(roots._h, mm._delta))
            mm._bound = roots

    def _traverse(roots, pb_c_base, pb_c_init, discount_factor, min_max_stats_lst, results, virtual_to_play_batch):
        if roots._h is None:
            raise L.LzError("batch_traverse before Roots.prepare")
        _bind(roots, min_max_stats_lst)
        B = roots.root_num
        vtp = L.i32(virtual_to_play_batch).copy()
        if vtp.shape != (B,):
            raise ValueError("virtual_to_play_batch must have root_num entries")
        ix = np.zeros(B, np.int32); iy = np.zeros(B, np.int32); la = np.zeros(B, np.int32); sl = np.zeros(B, np.int32)
        L.check(L.lib().lz_batch_traverse(roots._h, int(pb_c_base), float(pb_c_init), float(discount_factor), vtp,
                                          ix, iy, la, sl))
        results._search_lens = sl.tolist()
        results._roots = roots
        return ix.tolist(), iy.tolist(), la.tolist(), vtp.tolist()

    if has_deterministic_flag:
        def batch_traverse(roots, pb_c_base, pb_c_init, discount_factor, min_max_stats_lst, results,
                           virtual_to_play_batch, deterministic=False):
            if roots._h is not None and roots._tiebreak is None:
                L.check(L.lib().lz_roots_set_tiebreak(roots._h, 0 if deterministic else 1, roots._seed))
            return _traverse(roots, pb_c_base, pb_c_init, discount_factor, min_max_stats_lst, results,
                             virtual_to_play_batch)

        def batch_backpropagate(current_latent_state_index, discount_factor, rewards, values, policies,
                                min_max_stats_lst, results, to_play_batch):
            roots = results._roots
            L.check(L.lib().lz_batch_backpropagate(roots._h, int(current_latent_state_index), float(discount_factor),
                                                   L.f32(rewards), L.f32(values), L.f32(policies), None,
                                                   L.i32(to_play_batch)))
    else:
        def batch_traverse(roots, pb_c_base, pb_c_init, discount_factor, min_max_stats_lst, results,
                           virtual_to_play_batch):
            return _traverse(roots, pb_c_base, pb_c_init, discount_factor, min_max_stats_lst, results,
                             virtual_to_play_batch)

        def batch_backpropagate(current_latent_state_index, discount_factor, value_prefixs, values, policies,
                                min_max_stats_lst, results, is_reset_list, to_play_batch):
            roots = results._roots
            rst = L.i32(is_reset_list)
            L.check(L.lib().lz_batch_backpropagate(roots._h, int(current_latent_state_index), float(discount_factor),
                                                   L.f32(value_prefixs), L.f32(values), L.f32(policies),
                                                   rst.ctypes.data, L.i32(to_play_batch)))

    # ---- ReZero (ez_tree.pyx:94-121, mz_tree.pyx:84-110)
    def batch_traverse_with_reuse(roots, pb_c_base, pb_c_init, discount_factor, min_max_stats_lst, results,
                                  virtual_to_play_batch, true_action, reuse_value):
        if roots._h is None:
            raise L.LzError("batch_traverse_with_reuse before Roots.prepare")
        _bind(roots, min_max_stats_lst)
        B = roots.root_num
        vtp = L.i32(virtual_to_play_batch).copy()
        if vtp.shape != (B,) or len(true_action) != B or len(reuse_value) != B:
            raise ValueError("virtual_to_play_batch, true_action and reuse_value must have root_num entries")
        ix = np.zeros(B, np.int32); iy = np.zeros(B, np.int32); la = np.zeros(B, np.int32); sl = np.zeros(B, np.int32)
        L.check(L.lib().lz_batch_traverse_with_reuse(roots._h, int(pb_c_base), float(pb_c_init), float(discount_factor), vtp,
                                                     L.i32(true_action), L.f32(reuse_value), ix, iy, la, sl))
        results._search_lens = sl.tolist()
        results._roots = roots
        return ix.tolist(), iy.tolist(), la.tolist(), vtp.tolist()

    def _backprop_reuse(current_latent_state_index, discount_factor, value_prefixs, values, policies, results, is_reset_list,
                        to_play_batch, no_inference_lst, reuse_lst, reuse_value_lst):
        roots = results._roots
        n = len(value_prefixs)
        vp, v, pol = L.f32(value_prefixs if n else [0.0]), L.f32(values if n else [0.0]), L.f32(policies if n else [[0.0]])
        rst = L.i32(is_reset_list) if is_reset_list is not None else None
        L.check(L.lib().lz_batch_backpropagate_with_reuse(
            roots._h, int(current_latent_state_index), float(discount_factor), vp.ctypes.data, v.ctypes.data, pol.ctypes.data, n,
            rst.ctypes.data if rst is not None else None, L.i32(to_play_batch), L.i32(no_inference_lst), L.i32(reuse_lst),
            L.f32(reuse_value_lst)))

    if has_deterministic_flag:
        def batch_backpropagate_with_reuse(current_latent_state_index, discount_factor, value_prefixs, values, policies,
                                           min_max_stats_lst, results, to_play_batch, no_inference_lst, reuse_lst, reuse_value_lst):
            _backprop_reuse(current_latent_state_index, discount_factor, value_prefixs, values, policies, results, None,
                            to_play_batch, no_inference_lst, reuse_lst, reuse_value_lst)
    else:
        def batch_backpropagate_with_reuse(current_latent_state_index, discount_factor, value_prefixs, values, policies,
                                           min_max_stats_lst, results, is_reset_list, to_play_batch, no_inference_lst, reuse_lst,
                                           reuse_value_lst):
            _backprop_reuse(current_latent_state_index, discount_factor, value_prefixs, values, policies, results, is_reset_list,
                            to_play_batch, no_inference_lst, reuse_lst, reuse_value_lst)

    return dict(MinMaxStatsList=MinMaxStatsList, ResultsWrapper=ResultsWrapper, Roots=Roots,
                batch_traverse=batch_traverse, batch_backpropagate=batch_backpropagate,
                batch_traverse_with_reuse=batch_traverse_with_reuse,
                batch_backpropagate_with_reuse=batch_backpropagate_with_reuse)
