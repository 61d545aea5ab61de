// lz_tree.hip -- on-device MCTS tree kernels for gfx950 (MI355X).
//
// One 64-lane wavefront owns one root's tree for a whole kernel: lanes span the children of the
// current node (one coalesced 16-byte edge record per lane), selection is a wave arg-max, the
// softmax/mean-Q sums are replayed in the reference's order with v_readlane so that every float
// matches the CPU tree bit for bit, and the backup is a parallel path gather followed by a short
// scalar recurrence.  One launch advances ALL roots by one simulation.
//
// Reference semantics restated here (LightZero v0.2.0):
//   lzero/mcts/ctree/ctree_efficientzero/lib/cnode.cpp   expand :88-151, add_exploration_noise :153-171,
//       compute_mean_q :173-212, cbackpropagate :482-575, cselect_child :651-695, cucb_score :756-814,
//       cbatch_traverse :886-963, cbatch_backpropagate :577-601, get_distributions/values :389-419
//   lzero/mcts/ctree/ctree_muzero/lib/cnode.cpp          (reward instead of value prefix, no is_reset)
//   lzero/mcts/ctree/common_lib/cminimax.cpp             update :19-26, normalize :33-45
//
// Compiled with -ffp-contract=off: the reference is built for baseline x86-64 (no FMA), so
// `r + gamma * v` must stay a rounded multiply followed by a rounded add.  expf/logf come from
// lz_math.h (bit-identical to the host libm the reference links); division and sqrt are the
// correctly rounded HIP defaults.
#include <stdlib.h>

#include "lz_tree_dev.h"

namespace {

// ------------------------------------------------------------------------------------------------
// prepare: CRoots::prepare / prepare_no_noise  (cnode.cpp:325-360) = expand root + noise + visit_count += 1
// ------------------------------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(64) void k_prepare(lz_tree_dev t, float noise_w, const float *__restrict__ noises,
                                                int ragged, const int32_t *__restrict__ noise_off,
                                                const float *__restrict__ vps, const float *__restrict__ logits,
                                                const int32_t *__restrict__ to_play)
{
    extern __shared__ float sm[];  // [A] prior by action, then [A] int flag
    const int b = blockIdx.x, lane = threadIdx.x;
    const int A = t.A, NN = t.NN;
    float *s_prior = sm;
    int *s_flag = reinterpret_cast<int *>(sm + A);
    for (int a = lane; a < A; a += 64) { s_prior[a] = 0.0f; s_flag[a] = 0; }
    __syncthreads();
    const int n = uni(t.n_legal[b]);
    float lg[NC], e[NC];
    int act[NC];
    float m = LZ_FLOAT_MIN;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int j = c * 64 + lane;
        act[c] = (j < n) ? t.legal[(size_t)b * A + j] : 0;
        lg[c] = (j < n) ? logits[(size_t)b * A + act[c]] : LZ_FLOAT_MIN;
        m = fmaxf(m, lg[c]);
    }
    m = wave_max(m);  // policy_max (order independent); starts from FLOAT_MIN like cnode.cpp:123
#pragma unroll
    for (int c = 0; c < NC; ++c) e[c] = lz_expf(lg[c] - m);
    float sum = 0.0f;  // policy_sum accumulated in legal-list order (cnode.cpp:132-137)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int cnt = min(64, n - c * 64);
        for (int j = 0; j < cnt; ++j) sum += rl_f(e[c], j);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int j = c * 64 + lane;
        if (j < n) {
            float prior = e[c] / sum;
            if (noises) {  // add_exploration_noise cnode.cpp:163-170: noise indexed by position in the legal list
                const float nz = ragged ? noises[noise_off[b] + j] : noises[(size_t)b * A + j];
                prior = prior * (1 - noise_w) + nz * noise_w;
            }
            s_prior[act[c]] = prior;
            s_flag[act[c]] = 1;
        }
    }
    __syncthreads();
    for (int a = lane; a < A; a += 64) {
        const size_t o = ((size_t)b * NN + 0) * A + a;
        t.edge[o] = make_float4(s_prior[a], __int_as_float(0), 0.0f, 0.0f);
        t.child[o] = s_flag[a] ? -1 : -2;
    }
    if (lane == 0) {
        const size_t o = (size_t)b * NN;
        t.node_vp[o] = vps[b];
        t.node_reset[o] = 0;
        t.node_to_play[o] = to_play[b];
        t.node_best[o] = -1;
        t.node_bidx[o] = b;   // CNode::batch_index of the root (cnode.cpp:334)
        t.node_link[o] = 0;   // the root: no parent, depth 0
        t.root_visit[b] = 1;  // visit_count += 1 (cnode.cpp:341)
        t.root_vsum[b] = 0.0f;
        // every prepare starts a new env-step: the random streams keyed by the epoch (stochastic tie-breaks, device-side Dirichlet
        // noise) move on.  Nothing in this kernel reads it; the kernels that do are launched after it.
        if (b == 0 && t.rng_epoch) t.rng_epoch[0] += 1u;
    }
}

template <int NC, int VARIANT>
__global__ __launch_bounds__(64) void k_traverse(lz_tree_dev t, lz_traverse_args a, float delta_max,
                                                 const int32_t *__restrict__ vtp_in)
{
    const int b = blockIdx.x;
    const tview v = global_view(t, b);
    tscal<NC> sc;
    load_scalars<NC>(t, b, sc);
    if (a.fresh_minmax) {   // a fresh MinMaxStatsList per search (mcts_ctree.py:778-779) without a launch of its own
        sc.mn = LZ_FLOAT_MAX;
        sc.mx = LZ_FLOAT_MIN;
        if (threadIdx.x == 0) { t.minmax[2 * b] = LZ_FLOAT_MAX; t.minmax[2 * b + 1] = LZ_FLOAT_MIN; }
    }
    dev_traverse<NC, VARIANT>(t, v, sc, a, delta_max, vtp_in[b]);
}

template <int NC, int VARIANT>
__global__ __launch_bounds__(64) void k_backprop(lz_tree_dev t, int new_node, float discount,
                                                 const float *__restrict__ vps, const float *__restrict__ values,
                                                 const float *__restrict__ logits,
                                                 const int32_t *__restrict__ is_reset, int horizon,
                                                 const int32_t *__restrict__ to_play_in)
{
    const int b = blockIdx.x;
    const tview v = global_view(t, b);
    tscal<NC> sc;
    load_scalars<NC>(t, b, sc);
    leaf_in<NC, VARIANT> L;
    load_leaf<NC, VARIANT>(t, b, vps, values, logits, is_reset, horizon, to_play_in, L);
    dev_backprop<NC, VARIANT, false>(t, v, sc, new_node, discount, L.vp, L.value, L.lg, L.d, L.to_play, L.reset);
}

// expand + backup of simulation s followed by the selection of simulation s + 1 for the same root, in one launch
// (same wavefront owns the root in both phases; a workgroup-scope fence orders its stores before its loads).
template <int NC, int VARIANT>
__global__ __launch_bounds__(64) void k_backprop_traverse(lz_tree_dev t, int new_node, float discount,
                                                          const float *__restrict__ vps,
                                                          const float *__restrict__ values,
                                                          const float *__restrict__ logits, int horizon,
                                                          lz_traverse_args a, float delta_max,
                                                          const int32_t *__restrict__ vtp_in)
{
    const int b = blockIdx.x;
    const tview v = global_view(t, b);
    tscal<NC> sc;
    load_scalars<NC>(t, b, sc);
    leaf_in<NC, VARIANT> L;
    load_leaf<NC, VARIANT>(t, b, vps, values, logits, nullptr, horizon, nullptr, L);
    const int vtp = vtp_in[b];
#ifdef LZ_DEBUG_KNOBS
    const bool stamp = a.dbg_ts && b < 64 && threadIdx.x == 0;
#define LZ_SEP_TS(i) do { if (stamp) a.dbg_ts[b * 8 + i] = __builtin_readcyclecounter(); } while (0)
#else
#define LZ_SEP_TS(i) do { } while (0)
#endif
    LZ_SEP_TS(0);
    LZ_SEP_TS(1);
    dev_backprop<NC, VARIANT, false>(t, v, sc, new_node, discount, L.vp, L.value, L.lg, L.d, L.to_play, L.reset);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    LZ_SEP_TS(2);
    dev_traverse<NC, VARIANT>(t, v, sc, a, delta_max, vtp);
    LZ_SEP_TS(3);
#ifdef LZ_DEBUG_KNOBS
    if (stamp) { a.dbg_ts[b * 8 + 4] = (unsigned long long)L.d; a.dbg_ts[b * 8 + 5] = (unsigned long long)t.res_search_len[b]; }
#endif
#undef LZ_SEP_TS
}

// ReZero (search_with_reuse): cbatch_traverse_with_reuse / cbatch_backpropagate_with_reuse (cnode.cpp:603-649, 965-1072)
template <int NC, int VARIANT>
__global__ __launch_bounds__(64) void k_traverse_reuse(lz_tree_dev t, lz_traverse_args a, float delta_max,
                                                       const int32_t *__restrict__ vtp_in, const int32_t *__restrict__ true_action,
                                                       const float *__restrict__ reuse_value)
{
    const int b = blockIdx.x;
    const tview v = global_view(t, b);
    tscal<NC> sc;
    load_scalars<NC>(t, b, sc);
    dev_traverse<NC, VARIANT, true>(t, v, sc, a, delta_max, vtp_in[b], true_action[b], reuse_value[b]);
}

// mode[b]: 0 expand + back up the network value, 1 no inference (leaf already expanded; back up the reuse value),
// 2 expand + back up the reuse value; null => derived from the last traverse (fused path).  row[b]: row of this root in
// the packed network outputs (the reference packs the roots that needed inference); null => b.
template <int NC, int VARIANT>
__global__ __launch_bounds__(64) void k_backprop_reuse(lz_tree_dev t, int new_node, float discount,
                                                       const float *__restrict__ vps, const float *__restrict__ values,
                                                       const float *__restrict__ logits, const int32_t *__restrict__ is_reset,
                                                       int horizon, const int32_t *__restrict__ to_play_in,
                                                       const int32_t *__restrict__ mode, const int32_t *__restrict__ row,
                                                       const float *__restrict__ reuse_value,
                                                       const int32_t *__restrict__ true_action, int32_t *infer_counter)
{
    const int b = blockIdx.x, lane = threadIdx.x, A = t.A;
    const tview v = global_view(t, b);
    tscal<NC> sc;
    load_scalars<NC>(t, b, sc);
    const int d = uni(t.res_search_len[b]);
    int m;
    if (mode) m = uni(mode[b]);
    else m = uni(t.res_noinf[b]) ? 1 : ((uni(t.res_ix[b]) == 0 && uni(t.res_last_action[b]) == uni(true_action[b])) ? 2 : 0);
    const int r = row ? uni(row[b]) : b;
    const int to_play = uni(to_play_in ? to_play_in[b] : t.res_vtp[b]);
    int reset = 0;
    if (VARIANT == LZ_TREE_EFFICIENTZERO) {
        if (is_reset) reset = is_reset[b];
        else if (horizon > 0) reset = (d % horizon == 0) ? 1 : 0;
    }
    float lg[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int j = c * 64 + lane;
        lg[c] = (m != 1 && j < A) ? logits[(size_t)r * A + j] : LZ_FLOAT_MIN;
    }
    const float vp = (m != 1) ? vps[r] : 0.0f;
    const float value = (m != 0) ? reuse_value[b] : values[r];
    if (infer_counter && lane == 0 && m != 1) atomicAdd(infer_counter, 1);  // roots that used a network evaluation
    dev_backprop<NC, VARIANT, false>(t, v, sc, new_node, discount, vp, value, lg, d, to_play, reset, m == 1, r);
}

// The same fused step on an LDS copy of the root's tree (dev_step_lds).  Used when the tree of one root fits the LDS budget
// (lz_tree_lds_bytes).
template <int NC, int VARIANT>
__global__ __launch_bounds__(64) void k_backprop_traverse_lds(lz_tree_dev t, int new_node, float discount,
                                                              const float *__restrict__ vps,
                                                              const float *__restrict__ values,
                                                              const float *__restrict__ logits, int horizon,
                                                              lz_traverse_args a, float delta_max,
                                                              const int32_t *__restrict__ vtp_in)
{
    extern __shared__ __attribute__((aligned(16))) float4 s_tree[];
    dev_step_lds<NC, VARIANT>(t, blockIdx.x, new_node, discount, vps, values, logits, horizon, a, delta_max, vtp_in, s_tree);
}

// ------------------------------------------------------------------------------------------------
// k_tree_step_wg: the separate tree step (expand + backup of simulation s, selection of s + 1) for trees that have outgrown the LDS
// budget of the fused prologue -- BASELINE configs[2]: 1024 roots x 400 simulations, A = 4, paths ~50 deep.  k_backprop_traverse
// walks such a path with ONE wave of ~600 dependent instructions per level (2.4 k cycles x depth 48: 72 us per launch, 22 % of that
// configuration's GPU time, 4 of 64 lanes busy).  Here a workgroup of four waves owns the root:
//   0  wave 0: expand + backup on the HBM arrays (dev_backprop, unchanged);
//   1  all waves: EVERY expanded node at once (thread = node, as in dev_traverse_par): compute_mean_q's total / count in list order,
//      cucb_score of every visited child in full and the prior term of every unvisited one (all they lack is the node's mean_q, which
//      chains down the search path), and -- for a node whose children are all visited -- cselect_child itself; per node 48 bytes of
//      results in LDS (record, child scores, child ids);
//   2  wave 0: the walk -- per level one LDS record, mean_q = (parent_q + total) / (count + 1), then either the stored choice or
//      prior + normalised mean_q for the unvisited children and the arg-max over <= 8 lanes.
// Same float operations per node and level as dev_traverse, in the same order: bit-identical (the configs[2] exact-replay and
// fused-vs-separate tests run through it; LZ_TRAVERSE_SERIAL=1 keeps k_backprop_traverse).
template <int AU, int VARIANT>
__global__ __launch_bounds__(256) void k_tree_step_wg(lz_tree_dev t, int new_node, float discount, const float *__restrict__ vps,
                                                      const float *__restrict__ values, const float *__restrict__ logits, int horizon,
                                                      lz_traverse_args a, float delta_max, const int32_t *__restrict__ vtp_in)
{
    extern __shared__ __attribute__((aligned(16))) float4 s_wg[];
    const int b = blockIdx.x, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int A = t.A, NN = t.NN, nn = new_node + 1;
    // LDS: rec [nn] {total, count (int), sel_act (int; -1: not all children visited), sel_child (int)} | sc [nn][AU] | chd [nn][AU] | scalars
    float4 *s_rec = s_wg;
    float *s_sc = reinterpret_cast<float *>(s_rec + nn);
    int32_t *s_chd = reinterpret_cast<int32_t *>(s_sc + (size_t)nn * AU);
    int32_t *s_um = s_chd + (size_t)nn * AU;          // [nn] bit j: child j is in the list and unvisited; bits 8..15: children in the list
    float *s_f = reinterpret_cast<float *>(s_um + nn);   // [0] min [1] max [2] root value sum
    int32_t *s_i = reinterpret_cast<int32_t *>(s_f + 4); // [0] root visit count [1] n_root [2] epoch [3] child of the root's action 0
    const tview v = global_view(t, b);
    int vtp = 0;
    if (wv == 0) {
        tscal<1> sc;
        load_scalars<1>(t, b, sc);
        leaf_in<1, VARIANT> L;
        load_leaf<1, VARIANT>(t, b, vps, values, logits, nullptr, horizon, nullptr, L);
        vtp = vtp_in[b];
        dev_backprop<1, VARIANT, false>(t, v, sc, new_node, discount, L.vp, L.value, L.lg, L.d, L.to_play, L.reset);
        if (lane == 0) {
            s_f[0] = sc.mn; s_f[1] = sc.mx; s_f[2] = sc.root_vsum;
            s_i[0] = sc.root_visit; s_i[1] = sc.n_root; s_i[2] = (int32_t)sc.epoch;
        }
    }
    __syncthreads();   // the backup's stores (edges of the path, the new node) are visible to the workgroup; scalars in LDS
    const float mn = s_f[0], mx = s_f[1];
    const int root_visit = s_i[0], n_root = s_i[1];
    const uint32_t epoch = (uint32_t)s_i[2];
    const float mm_d = mx - mn;
    const bool mm_on = mm_d > 0;
    const float mm_den = (mm_d < delta_max) ? delta_max : mm_d;
    const float base = (float)a.pb_c_base;
    const uint64_t hkey = mix64(a.seed ^ ((uint64_t)epoch << 20) ^ (uint64_t)a.counter) ^ ((uint64_t)b << 12);
    typedef float v4f __attribute__((ext_vector_type(4)));
    // ---- 1: every expanded node
    for (int n0 = 0; n0 < nn; n0 += 256) {
        const bool nvalid = n0 + tid < nn;
        const int n = nvalid ? n0 + tid : nn - 1;
        const bool is_root = n == 0;
        const uint64_t link = t.node_link[(size_t)b * NN + n];
        const int parent = (int)(link >> 40), act_in = (int)((link >> 24) & 0xffffu), depth_n = (int)(link & 0xffffffu);
        const float node_vp = v.node_vp[n];
        const int node_reset = v.node_reset[n];
        const int cnt_n = is_root ? n_root : A;
        v4f e[AU];
        int chd[AU];
#pragma unroll
        for (int j = 0; j < AU; ++j) {
            const int lj = t.legal[(size_t)b * A + min(j, A - 1)];   // the root's children are its legal list (wave-uniform loads)
            const int aj = is_root ? ((j < n_root) ? lj : 0) : min(j, A - 1);
            e[j] = *reinterpret_cast<const v4f *>(v.edge + (size_t)n * A + aj);
            chd[j] = v.child[(size_t)n * A + aj];
        }
        const int chd_a0 = v.child[(size_t)n * A];
        const int in_vis_e = __float_as_int(v.edge[(size_t)parent * A + act_in].y);
        const int in_vis = is_root ? root_visit : in_vis_e;
        float prior[AU], val[AU], tr[AU];
        int vis[AU];
        float total = 0.0f;
        int nv = 0;
#pragma unroll
        for (int j = 0; j < AU; ++j) {
            prior[j] = e[j].x;
            vis[j] = __float_as_int(e[j].y);
            const float qv = e[j].z / (float)max(vis[j], 1);
            val[j] = (vis[j] == 0) ? 0.0f : qv;
            if (VARIANT == LZ_TREE_EFFICIENTZERO) {
                const float dv = e[j].w - node_vp;
                tr[j] = (node_reset == 1) ? e[j].w : dv;
            } else {
                tr[j] = e[j].w;
            }
            const float qsa = tr[j] + discount * val[j];
            const bool visited = j < cnt_n && vis[j] > 0;
            const float t2 = total + qsa;
            total = visited ? t2 : total;
            nv += visited ? 1 : 0;
        }
        const float N = (float)(in_vis - 1);
        const float pbc0 = lz_logf((N + base + 1) / base) + a.pb_c_init, sq = sqrtf(N);
        float score[AU];
        float best = -__builtin_inff();
        int um = 0;
#pragma unroll
        for (int j = 0; j < AU; ++j) {
            float pb_c = pbc0 * (sq / (float)(vis[j] + 1));
            const float prior_score = pb_c * prior[j];
            const float vq = (a.players == 1) ? tr[j] + discount * val[j] : tr[j] + discount * (-val[j]);
            const float nq = (vq - mn) / mm_den;
            float value_score = mm_on ? nq : vq;
            value_score = (value_score < 0) ? 0.0f : ((value_score > 1) ? 1.0f : value_score);
            const bool unvis = vis[j] == 0;
            const float ucb = unvis ? prior_score : prior_score + value_score;   // an unvisited child lacks its value term: the walk adds it
            score[j] = (j < cnt_n) ? ucb : -__builtin_inff();
            um |= (j < cnt_n && unvis) ? (1 << j) : 0;
            best = fmaxf(best, score[j]);
        }
        // a node whose listed children are all visited: cselect_child now (the walk only follows it)
        int pos = -1;
#pragma unroll
        for (int j = AU - 1; j >= 0; --j) pos = (score[j] == best) ? j : pos;
        const bool ok = pos >= 0 && best > LZ_FLOAT_MIN;
        if (a.tiebreak == LZ_TIE_RANDOM) {
            const float thr = best - 0.000001f;
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < AU; ++j) cnt += (j == pos || (j > pos && score[j] >= thr)) ? 1 : 0;
            const bool draw = ok && cnt > 1 && um == 0;
            if (__ballot(draw)) {
                const uint64_t h = mix64(hkey ^ (uint64_t)depth_n);
                int r = (int)(((h >> 32) * (uint64_t)cnt) >> 32);
                int pick = pos;
#pragma unroll
                for (int j = 0; j < AU; ++j) {
                    const bool member = j == pos || (j > pos && score[j] >= thr);
                    pick = (member && r == 0) ? j : pick;
                    r -= member ? 1 : 0;
                }
                pos = draw ? pick : pos;
            }
        }
        int sel_child = chd_a0, sel_pos = 0;
#pragma unroll
        for (int j = 0; j < AU; ++j)
            if (ok && pos == j) { sel_child = chd[j]; sel_pos = j; }
        if (nvalid) {
            // sel < 0: some listed child is unvisited, the walk decides.  (list position, not action: the root maps it through its list)
            const int sel = (um == 0) ? (ok ? sel_pos : 0x10000) : -1;    // 0x10000: no comparable score -> action 0 (cnode.cpp:687-693)
            float4 r4;
            r4.x = total; r4.y = __int_as_float(nv); r4.z = __int_as_float(sel); r4.w = __int_as_float(sel_child);
            s_rec[n] = r4;
            s_um[n] = um | (cnt_n << 8);
#pragma unroll
            for (int j = 0; j < AU; ++j) { s_sc[(size_t)n * AU + j] = score[j]; s_chd[(size_t)n * AU + j] = chd[j]; }
            if (is_root) s_i[3] = chd_a0;
        }
    }
    __syncthreads();
    if (wv != 0) return;
    // ---- 2: the walk (wave 0; lane j = child j of the current node where a node still has unvisited children)
    int node = 0, depth = 0, last_action = -1;
    int my_node = 0, my_act = 0;
    auto flush_path = [&](int base_k, int count) {
        if (lane < count) {
            t.path_node[(size_t)b * NN + base_k + lane] = my_node;
            t.path_act[(size_t)b * NN + base_k + lane] = my_act;
            t.node_best[(size_t)b * NN + my_node] = my_act;
        }
    };
    float parent_q = 0.0f;
    const int lact = (lane < A) ? t.legal[(size_t)b * A + lane] : 0;   // the root's list: position -> action
    for (;;) {
        const float4 r4 = s_rec[node];
        const float total = r4.x;
        const int nv = __float_as_int(r4.y), sel = __float_as_int(r4.z);
        const bool is_root = node == 0;
        float mean_q;
        if (is_root && nv > 0) mean_q = total / (float)nv;
        else mean_q = (parent_q + total) / (float)(nv + 1);
        parent_q = mean_q;
        int pos, nxt;
        bool ok;
        if (sel >= 0) {   // all children visited: chosen in phase 1
            ok = sel != 0x10000;
            pos = ok ? sel : 0;
            nxt = __float_as_int(r4.w);
        } else {
            const int umc = s_um[node], cnt_n = umc >> 8;
            const int jj = min(lane, AU - 1);
            const float scj = s_sc[(size_t)node * AU + jj];
            const int chj = s_chd[(size_t)node * AU + jj];
            float u = mm_on ? (mean_q - mn) / mm_den : mean_q;     // CMinMaxStats::normalize of the unvisited children's value term
            u = (u < 0) ? 0.0f : ((u > 1) ? 1.0f : u);
            const bool unvis = (umc >> jj) & 1;
            const float ucb = unvis ? scj + u : scj;
            const float score = (lane < cnt_n && lane < AU) ? ucb : -__builtin_inff();
            const float best = row0_max(score);
            uint64_t mask = __ballot(score == best);
            pos = mask ? __builtin_ctzll(mask) : -1;
            if (a.tiebreak == LZ_TIE_RANDOM && pos >= 0) {
                const float thr = best - 0.000001f;
                uint64_t mk = __ballot(lane == pos || (lane > pos && score >= thr));
                const int cnt = __builtin_popcountll(mk);
                if (cnt > 1) {
                    const uint64_t h = mix64(hkey ^ (uint64_t)depth);
                    const int r = (int)(((h >> 32) * (uint64_t)cnt) >> 32);
                    for (int q = 0; q < r; ++q) mk &= mk - 1;
                    pos = __builtin_ctzll(mk);
                }
            }
            ok = pos >= 0 && best > LZ_FLOAT_MIN;
            if (ok) nxt = rl_i(chj, pos);
            else { pos = 0; nxt = is_root ? s_i[3] : rl_i(chj, 0); }
        }
        const int action = ok ? (is_root ? rl_i(lact, pos) : pos) : 0;
        if (a.players > 1) vtp = (vtp == 1) ? 2 : 1;  // cnode.cpp:932-943
        if (lane == (depth & 63)) { my_node = node; my_act = action; }
        last_action = action;
        depth += 1;
        if ((depth & 63) == 0) flush_path(depth - 64, 64);
        if (nxt < 0 || depth >= nn) break;  // reached an unexpanded child: the leaf
        node = nxt;
    }
    flush_path((depth - 1) & ~63, depth - ((depth - 1) & ~63));
    if (lane == 0) {
        t.res_ix[b] = node;
        t.res_iy[b] = b;
        t.res_last_action[b] = last_action;
        t.res_search_len[b] = depth;
        t.res_vtp[b] = vtp;
    }
}
static inline size_t lz_tree_step_wg_lds(int A_unroll, int nn) { return (size_t)nn * (16 + 8 * (size_t)A_unroll + 4) + 64; }

// ------------------------------------------------------------------------------------------------
// Gumbel MuZero (lzero/mcts/ctree/ctree_gumbel_muzero/lib/cnode.cpp).  Lane = legal position of the current node.
// ------------------------------------------------------------------------------------------------
// sum of v over the lanes of `mask`, added in lane order (the reference accumulates in index order)
__device__ __forceinline__ float ordered_add(float acc, float v, uint64_t mask)
{
    while (mask) {
        const int j = __builtin_ctzll(mask);
        acc += rl_f(v, j);
        mask &= mask - 1;
    }
    return acc;
}

// csoftmax (cnode.cpp:903-928) over the first n positions: x[c] holds position c*64 + lane
template <int NC>
__device__ __forceinline__ void dev_csoftmax(float (&x)[NC], int n)
{
    const int lane = threadIdx.x;
    float m = -__builtin_inff();
#pragma unroll
    for (int c = 0; c < NC; ++c) if (c * 64 + lane < n) m = fmaxf(m, x[c]);
    m = wave_max(m);
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float e = lz_expf(x[c] - m);
        sum = ordered_add(sum, e, __ballot(c * 64 + lane < n));
    }
    const float ls = lz_logf(sum);
#pragma unroll
    for (int c = 0; c < NC; ++c) x[c] = lz_expf(x[c] - m - ls);
}

// qtransform_completed_by_mix_value (cnode.cpp:984-1037) with the header defaults (maxvisit_init 50, value_scale 0.1,
// rescale_values, epsilon 1e-8); CNode::get_q :181-197, compute_mixed_value :930-966, rescale_qvalues :968-982
// psoft = csoftmax of the node's child priors (compute_mixed_value's first step), cached per node in t.gsoft
template <int NC>
__device__ __forceinline__ void dev_completed_q(const float (&psoft)[NC], const int (&vis)[NC], const float (&q)[NC], int n, float raw_value,
                                                float (&cq)[NC])
{
    const int lane = threadIdx.x;
    float ptmp[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) ptmp[c] = psoft[c];
    float visit_count_sum = 0.0f, probs_sum = 0.0f, weighted_q_sum = 0.0f;
    const float min_num = -10e7f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const uint64_t valid = __ballot(c * 64 + lane < n);
        visit_count_sum = ordered_add(visit_count_sum, (float)vis[c], valid);
        ptmp[c] = fmaxf(ptmp[c], min_num);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) probs_sum = ordered_add(probs_sum, ptmp[c], __ballot(c * 64 + lane < n && vis[c] > 0));
#pragma unroll
    for (int c = 0; c < NC; ++c) weighted_q_sum = ordered_add(weighted_q_sum, ptmp[c] * q[c] / probs_sum, __ballot(c * 64 + lane < n && vis[c] > 0));
    const float value = (raw_value + visit_count_sum * weighted_q_sum) / (visit_count_sum + 1);
    float mx = -__builtin_inff(), mn = __builtin_inff(), max_visit = 0.0f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        cq[c] = vis[c] > 0 ? q[c] : value;
        if (c * 64 + lane < n) { mx = fmaxf(mx, cq[c]); mn = fminf(mn, cq[c]); max_visit = fmaxf(max_visit, (float)vis[c]); }
    }
    mx = wave_max(mx);
    mn = wave_min(mn);
    max_visit = wave_max(max_visit);
    const float gap = fmaxf(mx - mn, 1e-8f);
    const float visit_scale = 50.0f + max_visit;
#pragma unroll
    for (int c = 0; c < NC; ++c) cq[c] = (cq[c] - mn) / gap * visit_scale * 0.1f;
}

// children of `node` as seen by cselect_root_child / cselect_interior_child: prior, visit, q = reward + discount * value()
template <int NC>
__device__ __forceinline__ void dev_gchildren(const lz_tree_dev &t, int b, int node, int n, bool is_root, float discount, float (&prior)[NC],
                                              int (&vis)[NC], float (&q)[NC], int (&act)[NC], int (&chd)[NC])
{
    const int lane = threadIdx.x, A = t.A, NN = t.NN;
    const float4 *edge_b = t.edge + (size_t)b * NN * A;
    const int32_t *child_b = t.child + (size_t)b * NN * A;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int j = c * 64 + lane;
        const bool valid = j < n;
        act[c] = valid ? (is_root ? t.legal[(size_t)b * A + j] : j) : 0;
        const float4 e = valid ? edge_b[(size_t)node * A + act[c]] : make_float4(0.f, 0.f, 0.f, 0.f);
        chd[c] = valid ? child_b[(size_t)node * A + act[c]] : -1;
        prior[c] = e.x;
        vis[c] = __float_as_int(e.y);
        const float val = (vis[c] == 0) ? 0.0f : e.z / (float)vis[c];
        q[c] = e.w + discount * val;
    }
}

// csoftmax of the child priors of `node` by legal position -> t.gsoft (once per node: the priors never change after the
// expansion, the root's after its noise)
template <int NC>
__device__ __forceinline__ void dev_store_gsoft(const lz_tree_dev &t, int b, int node, int n, bool is_root)
{
    const int lane = threadIdx.x, A = t.A, NN = t.NN;
    float p[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int j = c * 64 + lane;
        const int a = (j < n) ? (is_root ? t.legal[(size_t)b * A + j] : j) : 0;
        p[c] = (j < n) ? t.edge[((size_t)b * NN + node) * A + a].x : 0.0f;
    }
    dev_csoftmax<NC>(p, n);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int j = c * 64 + lane;
        if (j < n) t.gsoft[((size_t)b * NN + node) * A + j] = p[c];
    }
}
template <int NC>
__device__ __forceinline__ void dev_load_gsoft(const lz_tree_dev &t, int b, int node, int n, float (&p)[NC])
{
    const int lane = threadIdx.x, A = t.A, NN = t.NN;
#pragma unroll
    for (int c = 0; c < NC; ++c) p[c] = t.gsoft[((size_t)b * NN + node) * A + min(c * 64 + lane, A - 1)];
}

// first position (legal-list order) whose score equals the wave maximum; all -inf => position 0 (cnode.cpp:724-733)
template <int NC>
__device__ __forceinline__ int dev_first_argmax(const float (&score)[NC])
{
    float best = -__builtin_inff();
#pragma unroll
    for (int c = 0; c < NC; ++c) best = fmaxf(best, score[c]);
    best = wave_max(best);
    if (!(best > -__builtin_inff())) return 0;
    int pos = -1;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const uint64_t mask = __ballot(score[c] == best);
        if (pos < 0 && mask) pos = c * 64 + __builtin_ctzll(mask);
    }
    return pos;
}

// cbatch_traverse (cnode.cpp:834-897)
template <int NC>
__device__ __forceinline__ void dev_gtraverse(const lz_tree_dev &t, float discount)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int A = t.A, NN = t.NN;
    int node = 0, depth = 0, is_root = 1, last_action = -1;
    for (;;) {
        const int n = is_root ? uni(t.n_legal[b]) : A;
        float prior[NC], q[NC], cq[NC], score[NC];
        int vis[NC], act[NC], chd[NC];
        float psoft[NC];
        dev_load_gsoft<NC>(t, b, node, n, psoft);
        dev_gchildren<NC>(t, b, node, n, is_root != 0, discount, prior, vis, q, act, chd);
        dev_completed_q<NC>(psoft, vis, q, n, t.node_raw[(size_t)b * NN + node], cq);
        if (is_root) {
            // cselect_root_child :701-745 + score_considered :1096-1131
            int sim_index = 0;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                uint64_t valid = __ballot(c * 64 + lane < n);
                while (valid) { const int j = __builtin_ctzll(valid); sim_index += rl_i(vis[c], j); valid &= valid - 1; }
            }
            const int considered_visit = t.considered[min(sim_index, NN - 1)];
            float max_logit = -__builtin_inff();
#pragma unroll
            for (int c = 0; c < NC; ++c) if (c * 64 + lane < n) max_logit = fmaxf(max_logit, prior[c]);
            max_logit = wave_max(max_logit);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int j = c * 64 + lane;
                const float g = t.gumbel[min(j, A - 1)];
                const float sc = fmaxf(-1e9f, g + (prior[c] - max_logit) + cq[c]);
                score[c] = (j < n && vis[c] == considered_visit) ? sc : -__builtin_inff();
            }
        } else {
            // cselect_interior_child :747-790
            float probs[NC];
            int vsum = 0;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                probs[c] = prior[c] + cq[c];
                uint64_t valid = __ballot(c * 64 + lane < n);
                while (valid) { const int j = __builtin_ctzll(valid); vsum += rl_i(vis[c], j); valid &= valid - 1; }
            }
            dev_csoftmax<NC>(probs, n);
#pragma unroll
            for (int c = 0; c < NC; ++c) score[c] = (c * 64 + lane < n) ? probs[c] - (float)vis[c] / (float)(1 + vsum) : -__builtin_inff();
        }
        const int pos = dev_first_argmax<NC>(score);
        int action = 0, nxt = -1;
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if ((pos >> 6) == c) { action = rl_i(act[c], pos & 63); nxt = rl_i(chd[c], pos & 63); }
        is_root = 0;
        if (lane == 0) {
            t.node_best[(size_t)b * NN + node] = action;
            t.path_node[(size_t)b * NN + depth] = node;
            t.path_act[(size_t)b * NN + depth] = action;
        }
        last_action = action;
        depth += 1;
        if (nxt < 0) break;
        node = nxt;
    }
    if (lane == 0) {
        t.res_ix[b] = node;
        t.res_iy[b] = b;
        t.res_last_action[b] = last_action;
        t.res_search_len[b] = depth;
        t.res_vtp[b] = -1;
    }
}

template <int NC>
__global__ __launch_bounds__(64) void k_gtraverse(lz_tree_dev t, float discount)
{
    dev_gtraverse<NC>(t, discount);
}

// cbatch_back_propagate (cnode.cpp:633-652): CNode::expand + cback_propagate == the MuZero expand / one-player backup, plus the raw value
template <int NC, bool THEN_TRAVERSE>
__global__ __launch_bounds__(64) void k_gbackprop(lz_tree_dev t, int new_node, float discount, const float *__restrict__ rewards,
                                                  const float *__restrict__ values, const float *__restrict__ logits)
{
    const int b = blockIdx.x;
    const tview v = global_view(t, b);
    tscal<NC> sc;
    load_scalars<NC>(t, b, sc);
    leaf_in<NC, LZ_TREE_MUZERO> L;
    load_leaf<NC, LZ_TREE_MUZERO>(t, b, rewards, values, logits, nullptr, 0, nullptr, L);
    if (threadIdx.x == 0) t.node_raw[(size_t)b * t.NN + new_node] = L.value;
    dev_backprop<NC, LZ_TREE_MUZERO, false>(t, v, sc, new_node, discount, L.vp, L.value, L.lg, L.d, -1, 0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    dev_store_gsoft<NC>(t, b, new_node, t.A, false);
    if (THEN_TRAVERSE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        dev_gtraverse<NC>(t, discount);
    }
}

// CRoots::prepare (cnode.cpp:418-455): the MuZero root expansion + raw value
__global__ void k_graw_root(lz_tree_dev t, const float *__restrict__ values)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < t.B) t.node_raw[(size_t)b * t.NN] = values[b];
}

template <int NC>
__global__ __launch_bounds__(64) void k_gsoft_root(lz_tree_dev t)
{
    const int b = blockIdx.x;
    dev_store_gsoft<NC>(t, b, 0, uni(t.n_legal[b]), true);
}

// CNode::get_policy :350-375 and CNode::get_children_value :309-338 of every root: [B][A] each (either may be null)
template <int NC>
__global__ __launch_bounds__(64) void k_gpolicies(lz_tree_dev t, float discount, float *__restrict__ policies, float *__restrict__ children_values)
{
    const int b = blockIdx.x, lane = threadIdx.x, A = t.A;
    const int n = uni(t.n_legal[b]);
    float prior[NC], q[NC], cq[NC];
    int vis[NC], act[NC], chd[NC];
    float psoft[NC];
    dev_load_gsoft<NC>(t, b, 0, n, psoft);
    dev_gchildren<NC>(t, b, 0, n, true, discount, prior, vis, q, act, chd);
    dev_completed_q<NC>(psoft, vis, q, n, t.node_raw[(size_t)b * t.NN], cq);
    if (children_values) {
        for (int a = lane; a < A; a += 64) children_values[(size_t)b * A + a] = -__builtin_inff();
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < NC; ++c) if (c * 64 + lane < n) children_values[(size_t)b * A + act[c]] = cq[c];
    }
    if (policies) {
        // probs over the whole action space: -inf for illegal actions, csoftmax over all A entries in action order
        extern __shared__ float s_p[];
        for (int a = lane; a < A; a += 64) s_p[a] = -__builtin_inff();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < NC; ++c) if (c * 64 + lane < n) s_p[act[c]] = prior[c] + cq[c];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        float x[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) x[c] = (c * 64 + lane < A) ? s_p[c * 64 + lane] : -__builtin_inff();
        dev_csoftmax<NC>(x, A);
#pragma unroll
        for (int c = 0; c < NC; ++c) if (c * 64 + lane < A) policies[(size_t)b * A + c * 64 + lane] = x[c];
    }
}

// bumps the RNG epoch once per prepare (stochastic tie-break streams differ between env-steps even when the
// whole search is replayed from a captured graph with identical kernel arguments)
__global__ void k_bump_epoch(lz_tree_dev t)
{
    if (threadIdx.x == 0 && blockIdx.x == 0 && t.rng_epoch) t.rng_epoch[0] += 1u;
}

__global__ void k_minmax_reset(lz_tree_dev t)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < t.B) { t.minmax[2 * i] = LZ_FLOAT_MAX; t.minmax[2 * i + 1] = LZ_FLOAT_MIN; }  // cminimax.cpp:6-10
}

// get_distributions / get_values (cnode.cpp:389-419)
__global__ void k_readout(lz_tree_dev t, int32_t *__restrict__ dist, int32_t *__restrict__ cnt,
                          float *__restrict__ values)
{
    const int b = blockIdx.x;
    const int A = t.A, NN = t.NN;
    const int n = t.n_legal[b];
    for (int j = threadIdx.x; j < A; j += blockDim.x) {
        int v = -1;
        if (j < n) v = __float_as_int(t.edge[((size_t)b * NN) * A + t.legal[(size_t)b * A + j]].y);
        dist[(size_t)b * A + j] = v;
    }
    if (threadIdx.x == 0) {
        if (cnt) cnt[b] = n;
        const int rv = t.root_visit[b];
        if (values) values[b] = (rv == 0) ? 0.0f : t.root_vsum[b] / (float)rv;
    }
}

}  // namespace

__global__ void lz_k_trajectories(lz_tree_dev t, int32_t *__restrict__ out, int stride)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= t.B) return;
    const int A = t.A, NN = t.NN;
    int node = 0, k = 0;
    int best = t.node_best[(size_t)b * NN + node];  // CNode::get_trajectory cnode.cpp:241-261
    while (best >= 0 && k < stride - 1) {
        out[(size_t)b * stride + k++] = best;
        node = t.child[((size_t)b * NN + node) * A + best];
        if (node < 0) break;
        best = t.node_best[(size_t)b * NN + node];
    }
    out[(size_t)b * stride + k] = -1;
}

static inline int nchunks(int A) { return (A + 63) / 64; }
// Beyond 256 actions a node's children no longer fit four register chunks per lane: lz_tree_wide.hip walks them in a loop.
// LZ_TREE_WIDE=1 (parity tests) sends every MuZero / EfficientZero tree through those kernels, so that they can be compared bit for
// bit with the register kernels on the action spaces both serve.
static inline bool use_wide(const lz_tree_dev &t)
{
    if (t.variant != LZ_TREE_EFFICIENTZERO && t.variant != LZ_TREE_MUZERO) return false;
    if (nchunks(t.A) > 4) return true;
    const char *v = getenv("LZ_TREE_WIDE");
    return v && *v && *v != '0';
}

// The exploration factors of a node with visit count n = lane (cnode.cpp:720-727: pb_c = log((N + base + 1) / base) + init, times sqrt(N)): the
// same 64 values for every root and every simulation of a search, so they are computed once per search here instead of in front of every
// tree step (dev_step_lds), where the software logf and its two table fetches sat in front of the staging of the tree
__global__ __launch_bounds__(64) void k_explore_tab(float *out, int pb_c_base, float pb_c_init)
{
    const float nf = (float)threadIdx.x;
    out[threadIdx.x] = lz_logf((nf + (float)pb_c_base + 1) / (float)pb_c_base) + pb_c_init;
    out[64 + threadIdx.x] = sqrtf(nf);
}
void lz_tree_launch_explore_tab(float *out, int pb_c_base, float pb_c_init, hipStream_t s)
{
    hipLaunchKernelGGL(k_explore_tab, dim3(1), dim3(64), 0, s, out, pb_c_base, pb_c_init);
}

void lz_tree_launch_minmax_reset(const lz_tree_dev &t, hipStream_t s)
{
    hipLaunchKernelGGL(k_minmax_reset, dim3((t.B + 255) / 256), dim3(256), 0, s, t);
}

void lz_tree_launch_prepare(const lz_tree_dev &t, float noise_w, const float *d_noises, int ragged,
                            const int32_t *d_noise_off, const float *d_vp, const float *d_logits,
                            const int32_t *d_to_play, hipStream_t s)
{
    // (the root expansion is the same for every tree variant: the Gumbel tree beyond 256 actions takes the chunk-loop kernel too)
    if (use_wide(t) || nchunks(t.A) > 4) { lz_tree_wide_launch_prepare(t, noise_w, d_noises, ragged, d_noise_off, d_vp, d_logits, d_to_play, s); return; }
    const size_t sh = (size_t)t.A * 8;
    switch (nchunks(t.A)) {
    case 1: hipLaunchKernelGGL(k_prepare<1>, dim3(t.B), dim3(64), sh, s, t, noise_w, d_noises, ragged, d_noise_off, d_vp, d_logits, d_to_play); break;
    case 2: hipLaunchKernelGGL(k_prepare<2>, dim3(t.B), dim3(64), sh, s, t, noise_w, d_noises, ragged, d_noise_off, d_vp, d_logits, d_to_play); break;
    default: hipLaunchKernelGGL(k_prepare<4>, dim3(t.B), dim3(64), sh, s, t, noise_w, d_noises, ragged, d_noise_off, d_vp, d_logits, d_to_play); break;
    }
}

template <int V>
static void launch_traverse_v(const lz_tree_dev &t, const lz_traverse_args &a, float delta, const int32_t *vtp, hipStream_t s)
{
    switch (nchunks(t.A)) {
    case 1: hipLaunchKernelGGL((k_traverse<1, V>), dim3(t.B), dim3(64), 0, s, t, a, delta, vtp); break;
    case 2: hipLaunchKernelGGL((k_traverse<2, V>), dim3(t.B), dim3(64), 0, s, t, a, delta, vtp); break;
    default: hipLaunchKernelGGL((k_traverse<4, V>), dim3(t.B), dim3(64), 0, s, t, a, delta, vtp); break;
    }
}

void lz_tree_launch_traverse(const lz_tree_dev &t, const lz_traverse_args &a, float delta, const int32_t *d_vtp_in,
                             hipStream_t s)
{
    if (use_wide(t)) { lz_tree_wide_launch_traverse(t, a, delta, d_vtp_in, s); return; }   // lz_tree_wide.hip: children walked in a loop
    if (t.variant == LZ_TREE_EFFICIENTZERO) launch_traverse_v<LZ_TREE_EFFICIENTZERO>(t, a, delta, d_vtp_in, s);
    else launch_traverse_v<LZ_TREE_MUZERO>(t, a, delta, d_vtp_in, s);
}

template <int V>
static void launch_backprop_v(const lz_tree_dev &t, int idx, float discount, const float *vp, const float *val,
                              const float *lg, const int32_t *rst, int horizon, const int32_t *tp, hipStream_t s)
{
    switch (nchunks(t.A)) {
    case 1: hipLaunchKernelGGL((k_backprop<1, V>), dim3(t.B), dim3(64), 0, s, t, idx, discount, vp, val, lg, rst, horizon, tp); break;
    case 2: hipLaunchKernelGGL((k_backprop<2, V>), dim3(t.B), dim3(64), 0, s, t, idx, discount, vp, val, lg, rst, horizon, tp); break;
    default: hipLaunchKernelGGL((k_backprop<4, V>), dim3(t.B), dim3(64), 0, s, t, idx, discount, vp, val, lg, rst, horizon, tp); break;
    }
}

void lz_tree_launch_backprop(const lz_tree_dev &t, int latent_index, float discount, const float *d_vp,
                             const float *d_values, const float *d_logits, const int32_t *d_is_reset, int horizon,
                             const int32_t *d_to_play, hipStream_t s)
{
    if (use_wide(t)) { lz_tree_wide_launch_backprop(t, latent_index, discount, d_vp, d_values, d_logits, d_is_reset, horizon, d_to_play, s); return; }
    if (t.variant == LZ_TREE_EFFICIENTZERO)
        launch_backprop_v<LZ_TREE_EFFICIENTZERO>(t, latent_index, discount, d_vp, d_values, d_logits, d_is_reset, horizon, d_to_play, s);
    else
        launch_backprop_v<LZ_TREE_MUZERO>(t, latent_index, discount, d_vp, d_values, d_logits, d_is_reset, horizon, d_to_play, s);
}


template <int V>
static void launch_bt_v(const lz_tree_dev &t, int idx, float discount, const float *vp, const float *val, const float *lg,
                        int horizon, const lz_traverse_args &a, float delta, const int32_t *vtp, hipStream_t s)
{
    // path lengths are bounded by the node count, so the previous path fits the same [nn] arrays
    const char *no_lds = getenv("LZ_TREE_NO_LDS");  // parity tests compare the two instantiations
    const size_t lds = lz_tree_lds_bytes(t, idx);
    // LZ_TREE_LDS_LIMIT_SEPARATE (experiments): a larger LDS budget for the separate launch than for the chain's fused prologue
    // (lz_chain_fusable keeps lz_tree_lds_limit).  Measured on BASELINE configs[2] (trees up to 40 KB): 78.5 ms per step with 64 KB
    // against 79.9 ms with the default -- staging a large tree costs what the walk's LDS hits save.  What bounds the walk (debug stamps,
    // tools/tree_sep_timing.py: 2.4 k cycles per level at depth 48) is instruction issue of ONE wave per SIMD, not memory: a one-level
    // lookahead (every child's children requested under the scoring) and a per-launch table of the visit-count factors -- both
    // bit-exact -- left the launch at 73.8 us (73.5 before) and were taken out again.
    static const size_t sep_limit = []() { const char *v = getenv("LZ_TREE_LDS_LIMIT_SEPARATE"); return v && *v ? (size_t)strtoul(v, nullptr, 0) : (size_t)0; }();
    if (!no_lds && lds <= std::max(lz_tree_lds_limit(16 * 1024), sep_limit) && lds <= 64 * 1024 && nchunks(t.A) <= 2) {
        if (nchunks(t.A) == 1) hipLaunchKernelGGL((k_backprop_traverse_lds<1, V>), dim3(t.B), dim3(64), lds, s, t, idx, discount, vp, val, lg, horizon, a, delta, vtp);
        else hipLaunchKernelGGL((k_backprop_traverse_lds<2, V>), dim3(t.B), dim3(64), lds, s, t, idx, discount, vp, val, lg, horizon, a, delta, vtp);
        return;
    }
    // deep trees with few actions (BASELINE configs[2]): a workgroup per root scores every node at once (k_tree_step_wg)
    if (t.A <= 8 && !a.serial && !getenv("LZ_TREE_NO_WG")) {
        const int AU = t.A <= 4 ? 4 : (t.A <= 6 ? 6 : 8);
        const size_t wl = lz_tree_step_wg_lds(AU, idx + 1);
        if (wl <= 64 * 1024) {
            if (AU == 4) hipLaunchKernelGGL((k_tree_step_wg<4, V>), dim3(t.B), dim3(256), wl, s, t, idx, discount, vp, val, lg, horizon, a, delta, vtp);
            else if (AU == 6) hipLaunchKernelGGL((k_tree_step_wg<6, V>), dim3(t.B), dim3(256), wl, s, t, idx, discount, vp, val, lg, horizon, a, delta, vtp);
            else hipLaunchKernelGGL((k_tree_step_wg<8, V>), dim3(t.B), dim3(256), wl, s, t, idx, discount, vp, val, lg, horizon, a, delta, vtp);
            return;
        }
    }
    switch (nchunks(t.A)) {
    case 1: hipLaunchKernelGGL((k_backprop_traverse<1, V>), dim3(t.B), dim3(64), 0, s, t, idx, discount, vp, val, lg, horizon, a, delta, vtp); break;
    case 2: hipLaunchKernelGGL((k_backprop_traverse<2, V>), dim3(t.B), dim3(64), 0, s, t, idx, discount, vp, val, lg, horizon, a, delta, vtp); break;
    default: hipLaunchKernelGGL((k_backprop_traverse<4, V>), dim3(t.B), dim3(64), 0, s, t, idx, discount, vp, val, lg, horizon, a, delta, vtp); break;
    }
}

void lz_tree_launch_backprop_traverse(const lz_tree_dev &t, int latent_index, float discount, const float *d_vp,
                                      const float *d_values, const float *d_logits, int horizon,
                                      const lz_traverse_args &a, float delta, const int32_t *d_vtp_in, hipStream_t s)
{
    if (use_wide(t)) { lz_tree_wide_launch_backprop_traverse(t, latent_index, discount, d_vp, d_values, d_logits, horizon, a, delta, d_vtp_in, s); return; }
    if (t.variant == LZ_TREE_EFFICIENTZERO) launch_bt_v<LZ_TREE_EFFICIENTZERO>(t, latent_index, discount, d_vp, d_values, d_logits, horizon, a, delta, d_vtp_in, s);
    else launch_bt_v<LZ_TREE_MUZERO>(t, latent_index, discount, d_vp, d_values, d_logits, horizon, a, delta, d_vtp_in, s);
}

template <int V>
static void launch_reuse_v(const lz_tree_dev &t, const lz_traverse_args &a, float delta, const int32_t *vtp, const int32_t *ta,
                           const float *rv, hipStream_t s)
{
    switch (nchunks(t.A)) {
    case 1: hipLaunchKernelGGL((k_traverse_reuse<1, V>), dim3(t.B), dim3(64), 0, s, t, a, delta, vtp, ta, rv); break;
    case 2: hipLaunchKernelGGL((k_traverse_reuse<2, V>), dim3(t.B), dim3(64), 0, s, t, a, delta, vtp, ta, rv); break;
    default: hipLaunchKernelGGL((k_traverse_reuse<4, V>), dim3(t.B), dim3(64), 0, s, t, a, delta, vtp, ta, rv); break;
    }
}
void lz_tree_launch_traverse_reuse(const lz_tree_dev &t, const lz_traverse_args &a, float delta, const int32_t *d_vtp_in,
                                   const int32_t *d_true_action, const float *d_reuse_value, hipStream_t s)
{
    if (use_wide(t)) { lz_tree_wide_launch_traverse_reuse(t, a, delta, d_vtp_in, d_true_action, d_reuse_value, s); return; }
    if (t.variant == LZ_TREE_EFFICIENTZERO) launch_reuse_v<LZ_TREE_EFFICIENTZERO>(t, a, delta, d_vtp_in, d_true_action, d_reuse_value, s);
    else launch_reuse_v<LZ_TREE_MUZERO>(t, a, delta, d_vtp_in, d_true_action, d_reuse_value, s);
}

template <int V>
static void launch_bpreuse_v(const lz_tree_dev &t, int idx, float discount, const float *vp, const float *val, const float *lg,
                             const int32_t *rst, int horizon, const int32_t *tp, const int32_t *mode, const int32_t *row,
                             const float *rv, const int32_t *ta, int32_t *ic, hipStream_t s)
{
    switch (nchunks(t.A)) {
    case 1: hipLaunchKernelGGL((k_backprop_reuse<1, V>), dim3(t.B), dim3(64), 0, s, t, idx, discount, vp, val, lg, rst, horizon, tp, mode, row, rv, ta, ic); break;
    case 2: hipLaunchKernelGGL((k_backprop_reuse<2, V>), dim3(t.B), dim3(64), 0, s, t, idx, discount, vp, val, lg, rst, horizon, tp, mode, row, rv, ta, ic); break;
    default: hipLaunchKernelGGL((k_backprop_reuse<4, V>), dim3(t.B), dim3(64), 0, s, t, idx, discount, vp, val, lg, rst, horizon, tp, mode, row, rv, ta, ic); break;
    }
}
void lz_tree_launch_backprop_reuse(const lz_tree_dev &t, int latent_index, float discount, const float *d_vp, const float *d_values,
                                   const float *d_logits, const int32_t *d_is_reset, int horizon, const int32_t *d_to_play,
                                   const int32_t *d_mode, const int32_t *d_row, const float *d_reuse_value,
                                   const int32_t *d_true_action, int32_t *d_infer_counter, hipStream_t s)
{
    if (use_wide(t)) {
        lz_tree_wide_launch_backprop_reuse(t, latent_index, discount, d_vp, d_values, d_logits, d_is_reset, horizon, d_to_play, d_mode, d_row, d_reuse_value,
                                           d_true_action, d_infer_counter, s);
        return;
    }
    if (t.variant == LZ_TREE_EFFICIENTZERO)
        launch_bpreuse_v<LZ_TREE_EFFICIENTZERO>(t, latent_index, discount, d_vp, d_values, d_logits, d_is_reset, horizon, d_to_play, d_mode, d_row, d_reuse_value, d_true_action, d_infer_counter, s);
    else
        launch_bpreuse_v<LZ_TREE_MUZERO>(t, latent_index, discount, d_vp, d_values, d_logits, d_is_reset, horizon, d_to_play, d_mode, d_row, d_reuse_value, d_true_action, d_infer_counter, s);
}

// The Gumbel tree's kernels keep a node's children in NC register chunks; beyond 256 actions they are instantiated with 8 / 16 chunks
// (512 / 1024 actions: ten per-child values per chunk still fit the register file of a one-wave workgroup)
#define LZ_G_DISPATCH(LAUNCH)                                       \
    {                                                               \
        const int nc_ = nchunks(t.A);                               \
        if (nc_ <= 1) { constexpr int NCV = 1; LAUNCH; }            \
        else if (nc_ <= 2) { constexpr int NCV = 2; LAUNCH; }       \
        else if (nc_ <= 4) { constexpr int NCV = 4; LAUNCH; }       \
        else if (nc_ <= 8) { constexpr int NCV = 8; LAUNCH; }       \
        else { constexpr int NCV = 16; LAUNCH; }                    \
    }
void lz_gtree_launch_prepare(const lz_tree_dev &t, float noise_w, const float *d_noises, int ragged, const int32_t *d_noise_off,
                             const float *d_rewards, const float *d_values, const float *d_logits, const int32_t *d_to_play, hipStream_t s)
{
    lz_tree_launch_prepare(t, noise_w, d_noises, ragged, d_noise_off, d_rewards, d_logits, d_to_play, s);
    hipLaunchKernelGGL(k_graw_root, dim3((t.B + 255) / 256), dim3(256), 0, s, t, d_values);
    LZ_G_DISPATCH(hipLaunchKernelGGL((k_gsoft_root<NCV>), dim3(t.B), dim3(64), 0, s, t))
}
void lz_gtree_launch_traverse(const lz_tree_dev &t, float discount, hipStream_t s)
{
    LZ_G_DISPATCH(hipLaunchKernelGGL(k_gtraverse<NCV>, dim3(t.B), dim3(64), 0, s, t, discount))
}
void lz_gtree_launch_backprop(const lz_tree_dev &t, int idx, float discount, const float *r, const float *v, const float *lg, hipStream_t s)
{
    LZ_G_DISPATCH(hipLaunchKernelGGL((k_gbackprop<NCV, false>), dim3(t.B), dim3(64), 0, s, t, idx, discount, r, v, lg))
}
void lz_gtree_launch_backprop_traverse(const lz_tree_dev &t, int idx, float discount, const float *r, const float *v, const float *lg,
                                       hipStream_t s)
{
    LZ_G_DISPATCH(hipLaunchKernelGGL((k_gbackprop<NCV, true>), dim3(t.B), dim3(64), 0, s, t, idx, discount, r, v, lg))
}
void lz_gtree_launch_policies(const lz_tree_dev &t, float discount, float *d_policies, float *d_children_values, hipStream_t s)
{
    const size_t sh = (size_t)t.A * 4;
    LZ_G_DISPATCH(hipLaunchKernelGGL(k_gpolicies<NCV>, dim3(t.B), dim3(64), sh, s, t, discount, d_policies, d_children_values))
}

void lz_tree_launch_bump_epoch(const lz_tree_dev &t, hipStream_t s)
{
    hipLaunchKernelGGL(k_bump_epoch, dim3(1), dim3(64), 0, s, t);
}

void lz_tree_launch_readout(const lz_tree_dev &t, int32_t *d_dist, int32_t *d_cnt, float *d_values, hipStream_t s)
{
    hipLaunchKernelGGL(k_readout, dim3(t.B), dim3(64), 0, s, t, d_dist, d_cnt, d_values);
}

void lz_tree_launch_trajectories(const lz_tree_dev &t, int32_t *d_out, int stride, hipStream_t s)
{
    hipLaunchKernelGGL(lz_k_trajectories, dim3((t.B + 63) / 64), dim3(64), 0, s, t, d_out, stride);
}
