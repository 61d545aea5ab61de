"""TEST INFRASTRUCTURE ONLY -- ctypes face of oracle/ctree_oracle.c with the reference's Cython
module surface (lzero/mcts/ctree/ctree_efficientzero/ez_tree.pyx:6-121,
ctree_muzero/mz_tree.pyx): ``Roots``, ``MinMaxStatsList``, ``ResultsWrapper``, ``batch_traverse``,
``batch_backpropagate``.  Two module-like namespaces are exported: ``ez_tree`` and ``mz_tree``.
"""
import ctypes
import os
import subprocess
import types

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libctree_oracle.so")


_SO_S = os.path.join(_HERE, "libctree_sampled_oracle.so")
_SO_G = os.path.join(_HERE, "libctree_gumbel_oracle.so")


def build(force=False):
    for so, src in ((_SO, "ctree_oracle.c"), (_SO_S, "ctree_sampled_oracle.c"), (_SO_G, "ctree_gumbel_oracle.c")):
        srcp = os.path.join(_HERE, src)
        if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(srcp):
            subprocess.run(["make", "-C", _HERE, "-B", os.path.basename(so)], check=True, stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        P = ctypes.c_void_p
        ip = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
        fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
        L.otree_create.restype = P
        L.otree_create.argtypes = [ctypes.c_int] * 4 + [ip, ip]
        L.otree_destroy.argtypes = [P]
        L.otree_set_tiebreak.argtypes = [P, ctypes.c_int]
        L.otree_set_delta.argtypes = [P, ctypes.c_float]
        L.otree_prepare.argtypes = [P, ctypes.c_float, P, fp, fp, ip]
        L.otree_traverse.argtypes = [P, ctypes.c_int, ctypes.c_float, ctypes.c_float, ip, ip, ip, ip, ip]
        L.otree_backpropagate.argtypes = [P, ctypes.c_int, ctypes.c_float, fp, fp, fp, ip, ip]
        L.otree_traverse_with_reuse.argtypes = [P, ctypes.c_int, ctypes.c_float, ctypes.c_float, ip, ip, fp, ip, ip, ip, ip]
        L.otree_backpropagate_with_reuse.argtypes = [P, ctypes.c_int, ctypes.c_float, fp, fp, fp, ip, ip, ip, ip, fp]
        L.otree_get_distributions.argtypes = [P, ip, ip]
        L.otree_get_values.argtypes = [P, fp]
        L.otree_get_minmax.argtypes = [P, fp]
        L.otree_get_trajectories.argtypes = [P, ip, ctypes.c_int]
        L.otree_probe.restype = ctypes.c_int
        L.otree_probe.argtypes = [P, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, ip, fp, ip]
        _lib = L
    return _lib


def _f32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32))


def _i32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.int32))


def _make(variant):
    class MinMaxStatsList(object):
        def __init__(self, num):
            self.num = num
            self.delta = 0.0

        def set_delta(self, value_delta_max):
            self.delta = float(value_delta_max)

    class ResultsWrapper(object):
        def __init__(self, num):
            self.num = num
            self.search_lens = []

        def get_search_len(self):
            return self.search_lens

    class Roots(object):
        def __init__(self, root_num, legal_actions_list, action_space_size=None, max_simulations=512):
            self.root_num = root_num
            self._legal = [list(map(int, l)) for l in legal_actions_list]
            self._A = action_space_size
            self._S = max_simulations
            self._h = None
            self._tiebreak = 0
            self._mm_bound = None

        @property
        def num(self):
            return self.root_num

        def _ensure(self, A):
            if self._h is None:
                if self._A is None:
                    self._A = A
                cnt = _i32([len(l) for l in self._legal])
                flat = _i32([a for l in self._legal for a in l] or [0])
                self._h = lib().otree_create(variant, self.root_num, self._A, self._S, flat, cnt)
                lib().otree_set_tiebreak(self._h, self._tiebreak)

        def set_tiebreak(self, mode):
            self._tiebreak = int(mode)
            if self._h is not None:
                lib().otree_set_tiebreak(self._h, self._tiebreak)

        def prepare(self, root_noise_weight, noises, value_prefix_pool, policy_logits_pool, to_play_batch):
            logits = _f32(policy_logits_pool)
            self._ensure(logits.shape[1])
            nz = _f32([x for row in noises for x in row] or [0.0])
            lib().otree_prepare(self._h, root_noise_weight, nz.ctypes.data, _f32(value_prefix_pool), logits,
                                _i32(to_play_batch))

        def prepare_no_noise(self, value_prefix_pool, policy_logits_pool, to_play_batch):
            logits = _f32(policy_logits_pool)
            self._ensure(logits.shape[1])
            lib().otree_prepare(self._h, 0.0, None, _f32(value_prefix_pool), logits, _i32(to_play_batch))

        def get_distributions(self):
            out = np.zeros((self.root_num, self._A), np.int32)
            cnt = np.zeros(self.root_num, np.int32)
            lib().otree_get_distributions(self._h, out, cnt)
            return [out[i, :cnt[i]].tolist() for i in range(self.root_num)]

        def get_values(self):
            out = np.zeros(self.root_num, np.float32)
            lib().otree_get_values(self._h, out)
            return out.tolist()

        def get_minmax(self):
            out = np.zeros((self.root_num, 2), np.float32)
            lib().otree_get_minmax(self._h, out)
            return out

        def probe(self, env, pb_c_base, pb_c_init, discount_factor, players=1):
            """diagnostic: the next selection of root ``env`` level by level, without touching the tree ->
            [(latent index of the node, {action: cucb_score}, action the deterministic rule picks), ...]"""
            cap = self._S + 2
            lat = np.zeros(cap, np.int32); act = np.zeros(cap, np.int32); sc = np.zeros((cap, self._A), np.float32)
            n = lib().otree_probe(self._h, int(env), int(pb_c_base), pb_c_init, discount_factor, int(players), cap, lat, sc, act)
            return [(int(lat[i]), {a: float(sc[i, a]) for a in range(self._A) if sc[i, a] > -999999.0}, int(act[i])) for i in range(n)]

        def get_trajectories(self):
            stride = self._S + 2
            out = np.zeros((self.root_num, stride), np.int32)
            lib().otree_get_trajectories(self._h, out, stride)
            res = []
            for i in range(self.root_num):
                row = out[i].tolist()
                res.append(row[:row.index(-1)])
            return res

        def clear(self):
            if self._h is not None:
                lib().otree_destroy(self._h)
                self._h = None

        def __del__(self):
            try:
                self.clear()
            except Exception:
                pass

    def batch_traverse(roots, pb_c_base, pb_c_init, discount_factor, min_max_stats_lst, results, virtual_to_play_batch,
                       deterministic=None):
        B = roots.num
        if roots._mm_bound is not min_max_stats_lst:
            lib().otree_set_delta(roots._h, min_max_stats_lst.delta)
            roots._mm_bound = min_max_stats_lst
        vtp = _i32(virtual_to_play_batch).copy()
        ix = np.zeros(B, np.int32); iy = np.zeros(B, np.int32); la = np.zeros(B, np.int32); sl = np.zeros(B, np.int32)
        lib().otree_traverse(roots._h, int(pb_c_base), pb_c_init, discount_factor, vtp, ix, iy, la, sl)
        results.search_lens = sl.tolist()
        results._roots = roots
        return ix.tolist(), iy.tolist(), la.tolist(), vtp.tolist()

    if variant == 0:
        def batch_backpropagate(current_latent_state_index, discount_factor, value_prefixs, values, policies,
                                min_max_stats_lst, results, is_reset_list, to_play_batch):
            roots = results._roots
            lib().otree_backpropagate(roots._h, current_latent_state_index, discount_factor, _f32(value_prefixs),
                                      _f32(values), _f32(policies), _i32(is_reset_list), _i32(to_play_batch))
    else:
        def batch_backpropagate(current_latent_state_index, discount_factor, rewards, values, policies,
                                min_max_stats_lst, results, to_play_batch):
            roots = results._roots
            lib().otree_backpropagate(roots._h, current_latent_state_index, discount_factor, _f32(rewards),
                                      _f32(values), _f32(policies), _i32([0] * roots.num), _i32(to_play_batch))

    # ReZero (ez_tree.pyx:94-121, mz_tree.pyx:84-110)
    def batch_traverse_with_reuse(roots, pb_c_base, pb_c_init, discount_factor, min_max_stats_lst, results,
                                  virtual_to_play_batch, true_action, reuse_value):
        B = roots.num
        if roots._mm_bound is not min_max_stats_lst:
            lib().otree_set_delta(roots._h, min_max_stats_lst.delta)
            roots._mm_bound = min_max_stats_lst
        vtp = _i32(virtual_to_play_batch).copy()
        ix = np.zeros(B, np.int32); iy = np.zeros(B, np.int32); la = np.zeros(B, np.int32); sl = np.zeros(B, np.int32)
        lib().otree_traverse_with_reuse(roots._h, int(pb_c_base), pb_c_init, discount_factor, vtp, _i32(true_action),
                                        _f32(reuse_value), ix, iy, la, sl)
        results.search_lens = sl.tolist()
        results._roots = roots
        return ix.tolist(), iy.tolist(), la.tolist(), vtp.tolist()

    def _bp_reuse(current_latent_state_index, discount_factor, value_prefixs, values, policies, results, is_reset_list,
                  to_play_batch, no_inference_lst, reuse_lst, reuse_value_lst):
        roots = results._roots
        A = roots._A
        pol = _f32(policies).reshape(-1, A) if len(policies) else np.zeros((1, A), np.float32)
        lib().otree_backpropagate_with_reuse(roots._h, current_latent_state_index, discount_factor,
                                             _f32(value_prefixs if len(value_prefixs) else [0.0]),
                                             _f32(values if len(values) else [0.0]), pol, _i32(is_reset_list),
                                             _i32(to_play_batch), _i32(no_inference_lst), _i32(reuse_lst), _f32(reuse_value_lst))

    if variant == 0:
        def batch_backpropagate_with_reuse(current_latent_state_index, discount_factor, value_prefixs, values, policies,
                                           min_max_stats_lst, results, is_reset_list, to_play_batch, no_inference_lst,
                                           reuse_lst, reuse_value_lst):
            _bp_reuse(current_latent_state_index, discount_factor, value_prefixs, values, policies, results, is_reset_list,
                      to_play_batch, no_inference_lst, reuse_lst, reuse_value_lst)
    else:
        def batch_backpropagate_with_reuse(current_latent_state_index, discount_factor, value_prefixs, values, policies,
                                           min_max_stats_lst, results, to_play_batch, no_inference_lst, reuse_lst,
                                           reuse_value_lst):
            _bp_reuse(current_latent_state_index, discount_factor, value_prefixs, values, policies, results,
                      [0] * results._roots.num, to_play_batch, no_inference_lst, reuse_lst, reuse_value_lst)

    ns = types.SimpleNamespace(MinMaxStatsList=MinMaxStatsList, ResultsWrapper=ResultsWrapper, Roots=Roots,
                               batch_traverse=batch_traverse, batch_backpropagate=batch_backpropagate,
                               batch_traverse_with_reuse=batch_traverse_with_reuse,
                               batch_backpropagate_with_reuse=batch_backpropagate_with_reuse)
    return ns


ez_tree = _make(0)
mz_tree = _make(1)


# ------------------------------------------------------------------------------------------------
# Sampled EfficientZero (continuous actions): ctypes face of oracle/ctree_sampled_oracle.c with the surface of
# lzero/mcts/ctree/ctree_sampled_efficientzero/ezs_tree.pyx
# ------------------------------------------------------------------------------------------------
_slib = None


def slib():
    global _slib
    if _slib is None:
        build()
        L = ctypes.CDLL(_SO_S)
        P = ctypes.c_void_p
        ip = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
        fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
        L.stree_create.restype = P
        L.stree_create.argtypes = [ctypes.c_int] * 4
        L.stree_create_discrete.restype = P
        L.stree_create_discrete.argtypes = [ctypes.c_int] * 4
        L.stree_destroy.argtypes = [P]
        L.stree_set_clock.argtypes = [P, ctypes.c_uint64]
        L.stree_set_tiebreak.argtypes = [P, ctypes.c_int]
        L.stree_set_delta.argtypes = [P, ctypes.c_float]
        L.stree_prepare.argtypes = [P, ctypes.c_float, P, fp, fp, ip, P]
        L.stree_traverse.argtypes = [P, ctypes.c_int, ctypes.c_float, ctypes.c_float, ip, ip, ip, fp, ip]
        L.stree_backpropagate.argtypes = [P, ctypes.c_int, ctypes.c_float, fp, fp, fp, ip, ip, P]
        L.stree_get_distributions.argtypes = [P, ip]
        L.stree_get_values.argtypes = [P, fp]
        L.stree_get_minmax.argtypes = [P, fp]
        L.stree_get_actions.argtypes = [P, ctypes.c_int, fp]
        _slib = L
    return _slib


def _make_sampled():
    class MinMaxStatsList(object):
        def __init__(self, num):
            self.num = num
            self.delta = 0.0

        def set_delta(self, value_delta_max):
            self.delta = float(value_delta_max)

    class ResultsWrapper(object):
        def __init__(self, num):
            self.num = num
            self.search_lens = []

        def get_search_len(self):
            return self.search_lens

    class Roots(object):
        def __init__(self, root_num, legal_actions_list, action_space_size, num_of_sampled_actions,
                     continuous_action_space=True, max_simulations=512):
            self.root_num, self.K, self._S = root_num, num_of_sampled_actions, max_simulations
            self.continuous = bool(continuous_action_space)
            if self.continuous:
                self.D = action_space_size
                self._h = slib().stree_create(root_num, self.D, self.K, max_simulations)
            else:  # an action is the float of its index (cnode.cpp:436-441); the policy is `action_space_size` logits
                self.D = 1
                self._h = slib().stree_create_discrete(root_num, action_space_size, self.K, max_simulations)
            self._mm_bound = None
            self.given = None  # optional injected samples for the next expand: [B][K][D]

        @property
        def num(self):
            return self.root_num

        def set_clock(self, c):
            slib().stree_set_clock(self._h, int(c))

        def set_tiebreak(self, mode):
            slib().stree_set_tiebreak(self._h, int(mode))

        def _given_ptr(self):
            if self.given is None:
                return None
            self._g = _f32(self.given)
            self.given = None
            return self._g.ctypes.data

        def prepare(self, root_noise_weight, noises, value_prefix_pool, policy_logits_pool, to_play_batch):
            nz = _f32(noises)
            slib().stree_prepare(self._h, root_noise_weight, nz.ctypes.data, _f32(value_prefix_pool),
                                 _f32(policy_logits_pool), _i32(to_play_batch), self._given_ptr())

        def prepare_no_noise(self, value_prefix_pool, policy_logits_pool, to_play_batch):
            slib().stree_prepare(self._h, 0.0, None, _f32(value_prefix_pool), _f32(policy_logits_pool),
                                 _i32(to_play_batch), self._given_ptr())

        def get_distributions(self):
            out = np.zeros((self.root_num, self.K), np.int32)
            slib().stree_get_distributions(self._h, out)
            return out.tolist()

        def get_values(self):
            out = np.zeros(self.root_num, np.float32)
            slib().stree_get_values(self._h, out)
            return out.tolist()

        def get_minmax(self):
            out = np.zeros((self.root_num, 2), np.float32)
            slib().stree_get_minmax(self._h, out)
            return out

        def get_sampled_actions(self, record=0):
            out = np.zeros((self.root_num, self.K, self.D), np.float32)
            slib().stree_get_actions(self._h, record, out)
            return out.tolist()

        def clear(self):
            if self._h is not None:
                slib().stree_destroy(self._h)
                self._h = None

        def __del__(self):
            try:
                self.clear()
            except Exception:
                pass

    def batch_traverse(roots, pb_c_base, pb_c_init, discount_factor, min_max_stats_lst, results, virtual_to_play_batch,
                       continuous_action_space=True):
        B = roots.num
        if roots._mm_bound is not min_max_stats_lst:
            slib().stree_set_delta(roots._h, min_max_stats_lst.delta)
            roots._mm_bound = min_max_stats_lst
        vtp = _i32(virtual_to_play_batch).copy()
        ix = np.zeros(B, np.int32); iy = np.zeros(B, np.int32); sl = np.zeros(B, np.int32)
        la = np.zeros((B, roots.D), np.float32)
        slib().stree_traverse(roots._h, int(pb_c_base), pb_c_init, discount_factor, vtp, ix, iy, la, sl)
        results.search_lens = sl.tolist()
        results._roots = roots
        return ix.tolist(), iy.tolist(), la.tolist(), vtp.tolist()

    def batch_backpropagate(current_latent_state_index, discount_factor, value_prefixs, values, policies,
                            min_max_stats_lst, results, is_reset_list, to_play_batch):
        roots = results._roots
        slib().stree_backpropagate(roots._h, current_latent_state_index, discount_factor, _f32(value_prefixs),
                                   _f32(values), _f32(policies), _i32(is_reset_list), _i32(to_play_batch),
                                   roots._given_ptr())

    return types.SimpleNamespace(MinMaxStatsList=MinMaxStatsList, ResultsWrapper=ResultsWrapper, Roots=Roots,
                                 batch_traverse=batch_traverse, batch_backpropagate=batch_backpropagate)


ezs_tree = _make_sampled()


# ------------------------------------------------------------------------------------------------
# Gumbel MuZero: ctypes face of oracle/ctree_gumbel_oracle.c with the surface of
# lzero/mcts/ctree/ctree_gumbel_muzero/gmz_tree.pyx
# ------------------------------------------------------------------------------------------------
_glib = None


def glib():
    global _glib
    if _glib is None:
        build()
        L = ctypes.CDLL(_SO_G)
        P = ctypes.c_void_p
        ip = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
        fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
        L.gtree_create.restype = P
        L.gtree_create.argtypes = [ctypes.c_int] * 3 + [ip, ip]
        L.gtree_destroy.argtypes = [P]
        L.gtree_set_delta.argtypes = [P, ctypes.c_float]
        L.gtree_prepare.argtypes = [P, ctypes.c_float, P, fp, fp, fp, ip]
        L.gtree_traverse.argtypes = [P, ctypes.c_int, ctypes.c_int, ctypes.c_float, ip, ip, ip, ip, ip]
        L.gtree_backpropagate.argtypes = [P, ctypes.c_int, ctypes.c_float, fp, fp, fp, ip]
        L.gtree_get_distributions.argtypes = [P, ip, ip]
        L.gtree_get_values.argtypes = [P, fp]
        L.gtree_get_children_values.argtypes = [P, ctypes.c_float, fp]
        L.gtree_get_policies.argtypes = [P, ctypes.c_float, fp]
        L.gtree_get_trajectories.argtypes = [P, ip, ctypes.c_int]
        L.gtree_generate_gumbel.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_int, fp]
        L.gtree_considered_visits.argtypes = [ctypes.c_int, ctypes.c_int, ip]
        L.gtree_softmax.argtypes = [fp, ctypes.c_int]
        _glib = L
    return _glib


def _make_gumbel():
    class MinMaxStatsList(object):
        def __init__(self, num):
            self.num = num
            self.delta = 0.0

        def set_delta(self, value_delta_max):
            self.delta = float(value_delta_max)

    class ResultsWrapper(object):
        def __init__(self, num):
            self.num = num
            self.search_lens = []

        def get_search_len(self):
            return self.search_lens

    class Roots(object):
        def __init__(self, root_num, legal_actions_list, action_space_size=None, max_simulations=512):
            self.root_num = root_num
            self._legal = [list(map(int, l)) for l in legal_actions_list]
            self._A = action_space_size
            self._S = max_simulations
            self._h = None

        @property
        def num(self):
            return self.root_num

        def _ensure(self, A):
            if self._h is None:
                if self._A is None:
                    self._A = A
                cnt = _i32([len(l) for l in self._legal])
                flat = _i32([a for l in self._legal for a in l] or [0])
                self._h = glib().gtree_create(self.root_num, self._A, self._S, flat, cnt)

        def prepare(self, root_noise_weight, noises, value_prefix_pool, value_pool, policy_logits_pool, to_play_batch):
            logits = _f32(policy_logits_pool)
            self._ensure(logits.shape[1])
            nz = _f32([x for row in noises for x in row] or [0.0])
            glib().gtree_prepare(self._h, root_noise_weight, nz.ctypes.data, _f32(value_prefix_pool), _f32(value_pool), logits,
                                 _i32(to_play_batch))

        def prepare_no_noise(self, value_prefix_pool, value_pool, policy_logits_pool, to_play_batch):
            logits = _f32(policy_logits_pool)
            self._ensure(logits.shape[1])
            glib().gtree_prepare(self._h, 0.0, None, _f32(value_prefix_pool), _f32(value_pool), logits, _i32(to_play_batch))

        def get_distributions(self):
            out = np.zeros((self.root_num, self._A), np.int32)
            cnt = np.zeros(self.root_num, np.int32)
            glib().gtree_get_distributions(self._h, out, cnt)
            return [out[i, :cnt[i]].tolist() for i in range(self.root_num)]

        def get_values(self):
            out = np.zeros(self.root_num, np.float32)
            glib().gtree_get_values(self._h, out)
            return out.tolist()

        def get_children_values(self, discount, action_space_size):
            out = np.zeros((self.root_num, self._A), np.float32)
            glib().gtree_get_children_values(self._h, discount, out)
            return out.tolist()

        def get_policies(self, discount, action_space_size):
            out = np.zeros((self.root_num, self._A), np.float32)
            glib().gtree_get_policies(self._h, discount, out)
            return out.tolist()

        def get_trajectories(self):
            stride = self._S + 2
            out = np.zeros((self.root_num, stride), np.int32)
            glib().gtree_get_trajectories(self._h, out, stride)
            return [row[:row.index(-1)] for row in out.tolist()]

        def clear(self):
            if self._h is not None:
                glib().gtree_destroy(self._h)
                self._h = None

        def __del__(self):
            try:
                self.clear()
            except Exception:
                pass

    def batch_traverse(roots, num_simulations, max_num_considered_actions, discount, results, virtual_to_play_batch):
        B = roots.num
        vtp = _i32(virtual_to_play_batch)
        ix = np.zeros(B, np.int32); iy = np.zeros(B, np.int32); la = np.zeros(B, np.int32); sl = np.zeros(B, np.int32)
        glib().gtree_traverse(roots._h, int(num_simulations), int(max_num_considered_actions), discount, vtp, ix, iy, la, sl)
        results.search_lens = sl.tolist()
        results._roots = roots
        return ix.tolist(), iy.tolist(), la.tolist(), vtp.tolist()

    def batch_back_propagate(current_latent_state_index, discount, value_prefixs, values, policies, min_max_stats_lst, results,
                             to_play_batch):
        roots = results._roots
        glib().gtree_set_delta(roots._h, min_max_stats_lst.delta)
        glib().gtree_backpropagate(roots._h, current_latent_state_index, discount, _f32(value_prefixs), _f32(values), _f32(policies),
                                   _i32(to_play_batch))

    return types.SimpleNamespace(MinMaxStatsList=MinMaxStatsList, ResultsWrapper=ResultsWrapper, Roots=Roots,
                                 batch_traverse=batch_traverse, batch_back_propagate=batch_back_propagate)


gmz_tree = _make_gumbel()
