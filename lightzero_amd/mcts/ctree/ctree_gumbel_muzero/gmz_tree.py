"""Drop-in for lzero/mcts/ctree/ctree_gumbel_muzero/gmz_tree.pyx: ``Roots`` (prepare / prepare_no_noise with root rewards AND
values, get_distributions / get_values / get_trajectories / get_policies / get_children_values), ``MinMaxStatsList``,
``ResultsWrapper``, ``batch_traverse``, ``batch_back_propagate`` -- trees in HBM, HIP kernels (lightzero_amd/csrc/lz_tree.hip).
The tree is deterministic like the reference's (every node's Gumbel vector comes from std::mt19937(0), cnode.cpp:1133-1151)."""
import numpy as np

from .... import _lib as L
from .._tree_common import make_module as _make

_base = _make(3, has_deterministic_flag=True)
MinMaxStatsList = _base["MinMaxStatsList"]
ResultsWrapper = _base["ResultsWrapper"]


class Roots(_base["Roots"]):
    def prepare(self, root_noise_weight, noises, value_prefix_pool, value_pool, policy_logits_pool, to_play_batch):
        logits = L.f32(policy_logits_pool)
        if logits.ndim != 2 or logits.shape[0] != self.root_num:
            raise ValueError("policy_logits_pool must be [root_num][action_space_size]")
        self._ensure(logits.shape[1])
        nz = L.f32([x for row in noises for x in row] or [0.0])
        L.check(L.lib().lz_groots_prepare(self._h, float(root_noise_weight), nz.ctypes.data, L.f32(value_prefix_pool),
                                          L.f32(value_pool), logits, L.i32(to_play_batch)))

    def prepare_no_noise(self, value_prefix_pool, value_pool, policy_logits_pool, to_play_batch):
        logits = L.f32(policy_logits_pool)
        if logits.ndim != 2 or logits.shape[0] != self.root_num:
            raise ValueError("policy_logits_pool must be [root_num][action_space_size]")
        self._ensure(logits.shape[1])
        L.check(L.lib().lz_groots_prepare(self._h, 0.0, None, L.f32(value_prefix_pool), L.f32(value_pool), logits,
                                          L.i32(to_play_batch)))

    def collect_rows(self, temperature, deterministic, d_rows_ptr, row_words, frame_floats, timestep=None, seed=None, policy_width=None,
                     d_obs_ptr=None, discount=0.997):
        """after a fused search: the env-step rows with the improved policy in the extra block and arg-max(improved policy over the
        legal actions) as the action (gumbel_muzero.py:591-592); returns (header [B, 8 + 3 A], root policy logits)"""
        from .._tree_common import collect_rows_ex
        return collect_rows_ex(self, self._A, temperature, deterministic, d_rows_ptr, row_words, frame_floats, discount=discount,
                               timestep=timestep, seed=seed, policy_width=policy_width, d_obs_ptr=d_obs_ptr)

    def get_policies(self, discount, action_space_size):
        out = np.zeros((self.root_num, self._A), np.float32)
        L.check(L.lib().lz_groots_get_policies(self._h, float(discount), out.ctypes.data, None))
        return out.tolist()

    def get_children_values(self, discount, action_space_size):
        out = np.zeros((self.root_num, self._A), np.float32)
        L.check(L.lib().lz_groots_get_policies(self._h, float(discount), None, out.ctypes.data))
        return out.tolist()


def batch_traverse(roots, num_simulations, max_num_considered_actions, discount, results, virtual_to_play_batch):
    if roots._h is None:
        raise L.LzError("batch_traverse before Roots.prepare")
    B = roots.root_num
    vtp = L.i32(virtual_to_play_batch).copy()
    ix = np.zeros(B, np.int32); iy = np.zeros(B, np.int32); la = np.zeros(B, np.int32); sl = np.zeros(B, np.int32)
    L.check(L.lib().lz_gbatch_traverse(roots._h, int(num_simulations), int(max_num_considered_actions), float(discount), vtp, ix, iy, la, sl))
    results._search_lens = sl.tolist()
    results._roots = roots
    return ix.tolist(), iy.tolist(), la.tolist(), vtp.tolist()


def batch_back_propagate(current_latent_state_index, discount, value_prefixs, values, policies, min_max_stats_lst, results,
                         to_play_batch):
    roots = results._roots
    L.check(L.lib().lz_gbatch_back_propagate(roots._h, int(current_latent_state_index), float(discount), L.f32(value_prefixs),
                                             L.f32(values), L.f32(policies)))
