// lz_tree_wide.hip -- the MuZero / EfficientZero tree for action spaces beyond 256 (Chinese chess: 2086 moves,
// zoo/board_games/chinese_chess/config/chinese_chess_muzero_bot_mode_config.py:33; Go 19x19: 362).
//
// The tree kernels of lz_tree.hip keep a node's children in at most four 64-lane register chunks per lane.  Here a node's
// children are walked in 64-lane chunks by a LOOP: nothing of a child stays in registers between two passes over the
// node, every pass recomputes what it needs from the child's 16-byte edge record (HBM / L2: a node of 2086 children is
// 33 KB).  Same arithmetic, same order as the reference and as dev_traverse / dev_backprop / k_prepare:
//   * prepare      CRoots::prepare / prepare_no_noise   cnode.cpp:325-360, expand cnode.cpp:88-151, noise :153-171
//   * traverse     cbatch_traverse cnode.cpp:886-963; compute_mean_q :173-212; cucb_score :756-814; cselect_child :651-695
//   * backprop     cbatch_backpropagate cnode.cpp:577-601 = expand (here) + cbackpropagate (dev_backprop, lz_tree_dev.h)
// One wavefront per root.  Order-sensitive sums (the softmax denominator over the actions, total_unsigned_q over the
// visited children) are accumulated in list order by v_readlane, as everywhere in the tree code; max / arg-max are
// order independent.  Built with -ffp-contract=off (lightzero_amd/build.py).
#include "lz_tree_dev.h"

namespace {

// The score pass of one node: what every lane needs to score child j of `node` again.
struct wnode {
    const float4 *edge;     // the node's A edge records
    const int32_t *legal;   // the root's legal list (root only)
    int n;                  // children in the list
    int is_root;
    float node_vp;
    int node_reset;
    float mean_q, pbc0, sq, mn, mx, delta_max, discount;
    int players;
    int arm_action;         // ReZero, at the root only: the trajectory's true action is scored by carm_score (cnode.cpp:697-754); -1: none
    float reuse_value;
};

// cucb_score (cnode.cpp:756-814) of the child at list position j; -inf beyond the list.  The expressions are dev_traverse's.
// Loads of a pass are issued U chunks at a time (one round trip per U chunks instead of one per chunk: a pass over 2086 children is 33
// chunks, and a loop that waits for every chunk's load before it requests the next is 33 round trips of ~1 us); the chunks of a group are
// then consumed in list order.
constexpr int WU = 8;
struct wedge { float4 e; int act; };
__device__ __forceinline__ void wide_load(const wnode &w, int c0, wedge (&g)[WU])
{
    const int lane = threadIdx.x;
    if (w.is_root) {
#pragma unroll
        for (int u = 0; u < WU; ++u) { const int j = c0 + u * 64 + lane; g[u].act = j < w.n ? w.legal[j] : 0; }
    } else {
#pragma unroll
        for (int u = 0; u < WU; ++u) { const int j = c0 + u * 64 + lane; g[u].act = j < w.n ? j : 0; }
    }
#pragma unroll
    for (int u = 0; u < WU; ++u) {
        const int j = c0 + u * 64 + lane;
        g[u].e = j < w.n ? w.edge[g[u].act] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

template <int VARIANT>
__device__ __forceinline__ float wide_score(const wnode &w, int j, const wedge &ge)
{
    if (j >= w.n) return -__builtin_inff();
    const int act = ge.act;
    const float4 e = ge.e;
    const float prior = e.x;
    const int vis = __float_as_int(e.y);
    const float val = (vis == 0) ? 0.0f : e.z / (float)vis;  // CNode::value cnode.cpp:223-239
    float tr;
    if (VARIANT == LZ_TREE_EFFICIENTZERO) {
        tr = e.w - w.node_vp;
        if (w.node_reset == 1) tr = e.w;
    } else {
        tr = e.w;
    }
    float pb_c = w.pbc0 * (w.sq / (float)(vis + 1));
    const float prior_score = pb_c * prior;
    const bool arm = act == w.arm_action;      // carm_score instead of cucb_score: the reuse value stands in for the child's, no prior term once visited
    const float vchild = arm ? w.reuse_value : val;
    float value_score;
    if (vis == 0) value_score = w.mean_q;
    else if (w.players == 1) value_score = tr + w.discount * vchild;
    else value_score = tr + w.discount * (-vchild);
    value_score = mm_normalize(value_score, w.mn, w.mx, w.delta_max);
    if (value_score < 0) value_score = 0;
    else if (value_score > 1) value_score = 1;
    float ucb = prior_score + value_score;
    if (arm && vis != 0) ucb = value_score;
    return ucb;
}

__device__ __forceinline__ int wave_min_i(int v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = min(v, __shfl_xor(v, o));
    return uni(v);
}

// REUSE (ReZero, cnode.cpp:697-754, 816-884, 965-1072): as in dev_traverse -- the walk stops right below the root when the true action
// is selected; res_noinf marks roots whose reached node is already expanded.
template <int VARIANT, bool REUSE = false>
__device__ __forceinline__ void dev_traverse_wide(const lz_tree_dev &t, const tview &v, const tscal<1> &sc, const lz_traverse_args &a,
                                                  float delta_max, int vtp, int true_action = -1, float reuse_value = 0.0f)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int A = t.A, NN = t.NN;
    const float discount = a.discount;
    const float base = (float)a.pb_c_base;
    const uint32_t epoch = sc.epoch;
    const int32_t *legal = t.legal + (size_t)b * A;
    int node = 0, depth = 0, is_root = 1, last_action = -1, noinf = 0;
    int node_visit = sc.root_visit;
    int my_node = 0, my_act = 0;   // the path record of level d stays in lane d & 63 (dev_traverse)
    auto flush_path = [&](int base_, int count) {
        if (lane < count) {
            t.path_node[(size_t)b * NN + base_ + lane] = my_node;
            t.path_act[(size_t)b * NN + base_ + lane] = my_act;
            t.node_best[(size_t)b * NN + my_node] = my_act;
        }
    };
    float parent_q = 0.0f;

    for (;;) {
        wnode w;
        w.edge = v.edge + (size_t)node * A;
        w.legal = legal;
        w.n = is_root ? sc.n_root : A;
        w.is_root = is_root;
        w.node_vp = v.node_vp[node];
        w.node_reset = v.node_reset[node];
        w.mn = sc.mn; w.mx = sc.mx; w.delta_max = delta_max; w.discount = discount; w.players = a.players;
        w.arm_action = (REUSE && is_root) ? true_action : -1;
        w.reuse_value = reuse_value;
        // ---- compute_mean_q (cnode.cpp:173-212): total_unsigned_q over the visited children in list order
        float total = 0.0f;
        int nv = 0;
        for (int g0 = 0; g0 < w.n; g0 += 64 * WU) {
            wedge g[WU];
            wide_load(w, g0, g);
#pragma unroll
            for (int u = 0; u < WU; ++u) {
                const int c0 = g0 + u * 64;
                if (c0 >= w.n) break;
                const bool valid = c0 + lane < w.n;
                const float4 e = g[u].e;
                const int vis = __float_as_int(e.y);
                const float val = (vis == 0) ? 0.0f : e.z / (float)vis;
                float tr;
                if (VARIANT == LZ_TREE_EFFICIENTZERO) {
                    tr = e.w - w.node_vp;
                    if (w.node_reset == 1) tr = e.w;
                } else {
                    tr = e.w;
                }
                const float qsa = tr + discount * val;
                uint64_t mask = __ballot(valid && vis > 0);
                while (mask) {
                    const int j2 = __builtin_ctzll(mask);
                    total += rl_f(qsa, j2);
                    nv += 1;
                    mask &= mask - 1;
                }
            }
        }
        float mean_q;
        if (is_root && nv > 0) mean_q = total / (float)nv;
        else mean_q = (parent_q + total) / (float)(nv + 1);
        is_root = 0;
        parent_q = mean_q;
        w.mean_q = mean_q;
        // ---- cucb_score of every child; the first arg-max in list order
        const float N = (float)(node_visit - 1);
        w.pbc0 = lz_logf((N + base + 1) / base) + a.pb_c_init;
        w.sq = sqrtf(N);
        float lbest = -__builtin_inff();
        int lpos = 0x7fffffff;
        for (int g0 = 0; g0 < w.n; g0 += 64 * WU) {
            wedge g[WU];
            wide_load(w, g0, g);
#pragma unroll
            for (int u = 0; u < WU; ++u) {
                const int j = g0 + u * 64 + lane;
                const float s = wide_score<VARIANT>(w, j, g[u]);
                if (s > lbest) { lbest = s; lpos = j; }   // strict: the lane keeps the FIRST of its equal maxima
            }
        }
        const float best = wave_max(lbest);
        int pos = -1;
        if (best > LZ_FLOAT_MIN) {
            pos = wave_min_i(lbest == best ? lpos : 0x7fffffff);
            if (a.tiebreak == LZ_TIE_RANDOM) {
                // tie list = [first arg-max] + later entries with score >= max - 1e-6 (cnode.cpp:675-685)
                const float thr = best - 0.000001f;
                int cnt = 0;
                for (int g0 = (pos & ~63); g0 < w.n; g0 += 64 * WU) {
                    wedge g[WU];
                    wide_load(w, g0, g);
#pragma unroll
                    for (int u = 0; u < WU; ++u) {
                        const int j = g0 + u * 64 + lane;
                        const float s = wide_score<VARIANT>(w, j, g[u]);
                        cnt += __builtin_popcountll(__ballot(j == pos || (j > pos && s >= thr)));
                    }
                }
                if (cnt > 1) {
                    const uint64_t h = mix64(mix64(a.seed ^ ((uint64_t)epoch << 20) ^ (uint64_t)a.counter) ^ ((uint64_t)b << 12) ^ (uint64_t)depth);
                    int r = (int)(((h >> 32) * (uint64_t)cnt) >> 32);  // uniform index in [0, cnt)
                    const int first = pos;   // (pos is rewritten by the pick below; membership is relative to the first arg-max)
                    for (int g0 = (first & ~63); g0 < w.n && r >= 0; g0 += 64 * WU) {
                        wedge g[WU];
                        wide_load(w, g0, g);
#pragma unroll
                        for (int u = 0; u < WU; ++u) {
                            const int c0 = g0 + u * 64, j = c0 + lane;
                            const float s = wide_score<VARIANT>(w, j, g[u]);
                            uint64_t mk = __ballot(j == first || (j > first && s >= thr));
                            const int pc = __builtin_popcountll(mk);
                            if (r >= 0 && r < pc) {
                                for (int q = 0; q < r; ++q) mk &= mk - 1;
                                pos = c0 + __builtin_ctzll(mk);
                                r = -1;
                            } else if (r >= 0) {
                                r -= pc;
                            }
                        }
                    }
                }
            }
        }
        // no child in the tie list (every score NaN, cnode.cpp:687-693): action 0
        const int action = (pos >= 0) ? (w.is_root ? uni(legal[pos]) : pos) : 0;
        const int nxt = uni(v.child[(size_t)node * A + action]);
        const int sel_visit = uni(__float_as_int(w.edge[action].y));
        if (a.players > 1) vtp = (vtp == 1) ? 2 : 1;  // cnode.cpp:932-943
        if (lane == (depth & 63)) { my_node = node; my_act = action; }
        last_action = action;
        depth += 1;
        if ((depth & 63) == 0) flush_path(depth - 64, 64);
        if (REUSE && w.is_root && action == true_action) { noinf = nxt >= 0 ? 1 : 0; break; }  // cnode.cpp:1041-1044
        if (nxt < 0) break;  // reached an unexpanded child: the leaf
        node = nxt;
        node_visit = sel_visit;
    }
    flush_path((depth - 1) & ~63, depth - ((depth - 1) & ~63));
    if (lane == 0) {
        t.res_ix[b] = node;
        t.res_iy[b] = REUSE ? (noinf ? b : t.node_bidx[(size_t)b * NN + node]) : b;
        t.res_last_action[b] = last_action;
        t.res_search_len[b] = depth;
        t.res_vtp[b] = vtp;
        if (REUSE) t.res_noinf[b] = noinf;
    }
}

// CNode::expand (cnode.cpp:88-151) of node `new_node` from its A policy logits: every action is legal below the root.
// Returns this lane's prior of action `lane` (chunk 0), which dev_backprop writes again with the same bits.
__device__ __forceinline__ float dev_expand_wide(const tview &v, int new_node, int A, const float *__restrict__ lg)
{
    const int lane = threadIdx.x;
    auto load = [&](int g0, float (&x)[WU]) {   // WU chunks of logits per round trip (LZ_FLOAT_MIN beyond the list)
#pragma unroll
        for (int u = 0; u < WU; ++u) { const int j = g0 + u * 64 + lane; x[u] = j < A ? lg[j] : LZ_FLOAT_MIN; }
    };
    float m = LZ_FLOAT_MIN;
    for (int g0 = 0; g0 < A; g0 += 64 * WU) {
        float x[WU];
        load(g0, x);
#pragma unroll
        for (int u = 0; u < WU; ++u) m = fmaxf(m, x[u]);
    }
    m = wave_max(m);
    float sum = 0.0f;   // policy_sum in action order (cnode.cpp:132-137)
    for (int g0 = 0; g0 < A; g0 += 64 * WU) {
        float x[WU];
        load(g0, x);
#pragma unroll
        for (int u = 0; u < WU; ++u) {
            const int cnt = min(64, A - (g0 + u * 64));   // <= 0 beyond the list
            const float e = lz_expf(x[u] - m);
            for (int j = 0; j < cnt; ++j) sum += rl_f(e, j);
        }
    }
    float pri0 = 0.0f;
    for (int g0 = 0; g0 < A; g0 += 64 * WU) {
        float x[WU];
        load(g0, x);
#pragma unroll
        for (int u = 0; u < WU; ++u) {
            const int j = g0 + u * 64 + lane;
            if (j < A) {
                const float p = lz_expf(x[u] - m) / sum;
                if (g0 + u == 0) pri0 = p;
                v.edge[(size_t)new_node * A + j] = make_float4(p, __int_as_float(0), 0.0f, 0.0f);
                v.child[(size_t)new_node * A + j] = -1;
            }
        }
    }
    return pri0;
}

struct wleaf {
    int d, to_play, reset;
    float vp, value;
};
template <int VARIANT>
__device__ __forceinline__ wleaf load_leaf_wide(const lz_tree_dev &t, int b, const float *__restrict__ vps, const float *__restrict__ values,
                                                const int32_t *__restrict__ is_reset, int horizon, const int32_t *__restrict__ to_play_in)
{
    wleaf L;
    L.d = uni(t.res_search_len[b]);
    L.to_play = uni(to_play_in ? to_play_in[b] : t.res_vtp[b]);
    L.vp = vps[b];
    L.value = values[b];
    L.reset = 0;
    if (VARIANT == LZ_TREE_EFFICIENTZERO) {
        if (is_reset) L.reset = is_reset[b];
        else if (horizon > 0) L.reset = (L.d % horizon == 0) ? 1 : 0;  // mcts_ctree.py:859
    }
    return L;
}

// no_expand (ReZero, cnode.cpp:626-630): the leaf is an already expanded node; bidx: the leaf's batch_index
template <int VARIANT>
__device__ __forceinline__ void dev_backprop_wide(const lz_tree_dev &t, const tview &v, tscal<1> &sc, int new_node, float discount,
                                                  const wleaf &L, const float *__restrict__ lg, bool no_expand = false, int bidx = -1)
{
    float pri[1] = {0.0f};
    if (!no_expand) pri[0] = dev_expand_wide(v, new_node, t.A, lg);
    const float unused[1] = {0.0f};
    // the first 64 edges are written once more by dev_backprop (same values); node records, the link, the backup along the path
    dev_backprop<1, VARIANT, false>(t, v, sc, new_node, discount, L.vp, L.value, unused, L.d, L.to_play, L.reset, no_expand, bidx, nullptr, pri);
}

// prepare: expand the root over its legal list + noise + visit_count += 1 (k_prepare of lz_tree.hip with the chunks in a loop)
__global__ __launch_bounds__(64) void k_prepare_wide(lz_tree_dev t, float noise_w, const float *__restrict__ noises, int ragged,
                                                     const int32_t *__restrict__ noise_off, const float *__restrict__ vps,
                                                     const float *__restrict__ logits, const int32_t *__restrict__ to_play)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int A = t.A, NN = t.NN;
    const int n = uni(t.n_legal[b]);
    const int32_t *legal = t.legal + (size_t)b * A;
    const float *lgb = logits + (size_t)b * A;
    float4 *edge0 = t.edge + (size_t)b * NN * A;
    int32_t *child0 = t.child + (size_t)b * NN * A;
    for (int a = lane; a < A; a += 64) {   // actions outside the legal list: prior 0, child -2
        edge0[a] = make_float4(0.0f, __int_as_float(0), 0.0f, 0.0f);
        child0[a] = -2;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    float m = LZ_FLOAT_MIN;   // policy_max starts from FLOAT_MIN (cnode.cpp:123)
    for (int c0 = 0; c0 < n; c0 += 64) m = fmaxf(m, (c0 + lane < n) ? lgb[legal[c0 + lane]] : LZ_FLOAT_MIN);
    m = wave_max(m);
    float sum = 0.0f;         // policy_sum in legal-list order (cnode.cpp:132-137)
    for (int c0 = 0; c0 < n; c0 += 64) {
        const float e = lz_expf(((c0 + lane < n) ? lgb[legal[c0 + lane]] : LZ_FLOAT_MIN) - m);
        const int cnt = min(64, n - c0);
        for (int j = 0; j < cnt; ++j) sum += rl_f(e, j);
    }
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int j = c0 + lane;
        if (j < n) {
            const int act = legal[j];
            float prior = lz_expf(lgb[act] - m) / sum;
            if (noises) {  // add_exploration_noise cnode.cpp:163-170: noise indexed by position in the legal list
                const float nz = ragged ? noises[noise_off[b] + j] : noises[(size_t)b * A + j];
                prior = prior * (1 - noise_w) + nz * noise_w;
            }
            edge0[act] = make_float4(prior, __int_as_float(0), 0.0f, 0.0f);
            child0[act] = -1;
        }
    }
    if (lane == 0) {
        const size_t o = (size_t)b * NN;
        t.node_vp[o] = vps[b];
        t.node_reset[o] = 0;
        t.node_to_play[o] = to_play[b];
        t.node_best[o] = -1;
        t.node_bidx[o] = b;
        t.node_link[o] = 0;
        t.root_visit[b] = 1;  // visit_count += 1 (cnode.cpp:341)
        t.root_vsum[b] = 0.0f;
        if (b == 0 && t.rng_epoch) t.rng_epoch[0] += 1u;
    }
}

template <int VARIANT>
__global__ __launch_bounds__(64) void k_traverse_wide(lz_tree_dev t, lz_traverse_args a, float delta_max, const int32_t *__restrict__ vtp_in)
{
    const int b = blockIdx.x;
    const tview v = global_view(t, b);
    tscal<1> sc;
    load_scalars<1>(t, b, sc);
    if (a.fresh_minmax) {
        sc.mn = LZ_FLOAT_MAX;
        sc.mx = LZ_FLOAT_MIN;
        if (threadIdx.x == 0) { t.minmax[2 * b] = LZ_FLOAT_MAX; t.minmax[2 * b + 1] = LZ_FLOAT_MIN; }
    }
    dev_traverse_wide<VARIANT>(t, v, sc, a, delta_max, vtp_in[b]);
}

template <int VARIANT>
__global__ __launch_bounds__(64) void k_backprop_wide(lz_tree_dev t, int new_node, float discount, const float *__restrict__ vps,
                                                      const float *__restrict__ values, const float *__restrict__ logits,
                                                      const int32_t *__restrict__ is_reset, int horizon, const int32_t *__restrict__ to_play_in)
{
    const int b = blockIdx.x;
    const tview v = global_view(t, b);
    tscal<1> sc;
    load_scalars<1>(t, b, sc);
    const wleaf L = load_leaf_wide<VARIANT>(t, b, vps, values, is_reset, horizon, to_play_in);
    dev_backprop_wide<VARIANT>(t, v, sc, new_node, discount, L, logits + (size_t)b * t.A);
}

// ReZero: cbatch_traverse_with_reuse / cbatch_backpropagate_with_reuse (cnode.cpp:603-649, 965-1072); k_traverse_reuse / k_backprop_reuse of lz_tree.hip
template <int VARIANT>
__global__ __launch_bounds__(64) void k_traverse_reuse_wide(lz_tree_dev t, lz_traverse_args a, float delta_max, const int32_t *__restrict__ vtp_in,
                                                            const int32_t *__restrict__ true_action, const float *__restrict__ reuse_value)
{
    const int b = blockIdx.x;
    const tview v = global_view(t, b);
    tscal<1> sc;
    load_scalars<1>(t, b, sc);
    dev_traverse_wide<VARIANT, true>(t, v, sc, a, delta_max, vtp_in[b], true_action[b], reuse_value[b]);
}

template <int VARIANT>
__global__ __launch_bounds__(64) void k_backprop_reuse_wide(lz_tree_dev t, int new_node, float discount, const float *__restrict__ vps,
                                                            const float *__restrict__ values, const float *__restrict__ logits,
                                                            const int32_t *__restrict__ is_reset, int horizon, const int32_t *__restrict__ to_play_in,
                                                            const int32_t *__restrict__ mode, const int32_t *__restrict__ row,
                                                            const float *__restrict__ reuse_value, const int32_t *__restrict__ true_action,
                                                            int32_t *infer_counter)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const tview v = global_view(t, b);
    tscal<1> sc;
    load_scalars<1>(t, b, sc);
    wleaf L;
    L.d = uni(t.res_search_len[b]);
    int m;   // 0 expand + back up the network value, 1 no inference (back up the reuse value), 2 expand + back up the reuse value
    if (mode) m = uni(mode[b]);
    else m = uni(t.res_noinf[b]) ? 1 : ((uni(t.res_ix[b]) == 0 && uni(t.res_last_action[b]) == uni(true_action[b])) ? 2 : 0);
    const int r = row ? uni(row[b]) : b;
    L.to_play = uni(to_play_in ? to_play_in[b] : t.res_vtp[b]);
    L.reset = 0;
    if (VARIANT == LZ_TREE_EFFICIENTZERO) {
        if (is_reset) L.reset = is_reset[b];
        else if (horizon > 0) L.reset = (L.d % horizon == 0) ? 1 : 0;
    }
    L.vp = (m != 1) ? vps[r] : 0.0f;
    L.value = (m != 0) ? reuse_value[b] : values[r];
    if (infer_counter && lane == 0 && m != 1) atomicAdd(infer_counter, 1);
    dev_backprop_wide<VARIANT>(t, v, sc, new_node, discount, L, logits + (size_t)r * t.A, m == 1, r);
}

// expand + backup of simulation s, then the selection of simulation s + 1, in one launch (k_backprop_traverse of lz_tree.hip)
template <int VARIANT>
__global__ __launch_bounds__(64) void k_backprop_traverse_wide(lz_tree_dev t, int new_node, float discount, const float *__restrict__ vps,
                                                               const float *__restrict__ values, const float *__restrict__ logits, int horizon,
                                                               lz_traverse_args a, float delta_max, const int32_t *__restrict__ vtp_in)
{
    const int b = blockIdx.x;
    const tview v = global_view(t, b);
    tscal<1> sc;
    load_scalars<1>(t, b, sc);
    const wleaf L = load_leaf_wide<VARIANT>(t, b, vps, values, nullptr, horizon, nullptr);
    const int vtp = vtp_in[b];
    dev_backprop_wide<VARIANT>(t, v, sc, new_node, discount, L, logits + (size_t)b * t.A);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    dev_traverse_wide<VARIANT>(t, v, sc, a, delta_max, vtp);
}

}  // namespace

void lz_tree_wide_launch_prepare(const lz_tree_dev &t, float noise_w, const float *d_noises, int ragged, const int32_t *d_noise_off,
                                 const float *d_vp, const float *d_logits, const int32_t *d_to_play, hipStream_t s)
{
    hipLaunchKernelGGL(k_prepare_wide, dim3(t.B), dim3(64), 0, s, t, noise_w, d_noises, ragged, d_noise_off, d_vp, d_logits, d_to_play);
}

void lz_tree_wide_launch_traverse(const lz_tree_dev &t, const lz_traverse_args &a, float delta, const int32_t *d_vtp_in, hipStream_t s)
{
    if (t.variant == LZ_TREE_EFFICIENTZERO) hipLaunchKernelGGL((k_traverse_wide<LZ_TREE_EFFICIENTZERO>), dim3(t.B), dim3(64), 0, s, t, a, delta, d_vtp_in);
    else hipLaunchKernelGGL((k_traverse_wide<LZ_TREE_MUZERO>), dim3(t.B), dim3(64), 0, s, t, a, delta, d_vtp_in);
}

void lz_tree_wide_launch_backprop(const lz_tree_dev &t, int idx, float discount, const float *vp, const float *val, const float *lg,
                                  const int32_t *rst, int horizon, const int32_t *tp, hipStream_t s)
{
    if (t.variant == LZ_TREE_EFFICIENTZERO)
        hipLaunchKernelGGL((k_backprop_wide<LZ_TREE_EFFICIENTZERO>), dim3(t.B), dim3(64), 0, s, t, idx, discount, vp, val, lg, rst, horizon, tp);
    else
        hipLaunchKernelGGL((k_backprop_wide<LZ_TREE_MUZERO>), dim3(t.B), dim3(64), 0, s, t, idx, discount, vp, val, lg, rst, horizon, tp);
}

void lz_tree_wide_launch_traverse_reuse(const lz_tree_dev &t, const lz_traverse_args &a, float delta, const int32_t *vtp, const int32_t *ta,
                                        const float *rv, hipStream_t s)
{
    if (t.variant == LZ_TREE_EFFICIENTZERO) hipLaunchKernelGGL((k_traverse_reuse_wide<LZ_TREE_EFFICIENTZERO>), dim3(t.B), dim3(64), 0, s, t, a, delta, vtp, ta, rv);
    else hipLaunchKernelGGL((k_traverse_reuse_wide<LZ_TREE_MUZERO>), dim3(t.B), dim3(64), 0, s, t, a, delta, vtp, ta, rv);
}

void lz_tree_wide_launch_backprop_reuse(const lz_tree_dev &t, int idx, float discount, const float *vp, const float *val, const float *lg,
                                        const int32_t *rst, int horizon, const int32_t *tp, const int32_t *mode, const int32_t *row,
                                        const float *rv, const int32_t *ta, int32_t *ic, hipStream_t s)
{
    if (t.variant == LZ_TREE_EFFICIENTZERO)
        hipLaunchKernelGGL((k_backprop_reuse_wide<LZ_TREE_EFFICIENTZERO>), dim3(t.B), dim3(64), 0, s, t, idx, discount, vp, val, lg, rst, horizon, tp, mode, row, rv, ta, ic);
    else
        hipLaunchKernelGGL((k_backprop_reuse_wide<LZ_TREE_MUZERO>), dim3(t.B), dim3(64), 0, s, t, idx, discount, vp, val, lg, rst, horizon, tp, mode, row, rv, ta, ic);
}

void lz_tree_wide_launch_backprop_traverse(const lz_tree_dev &t, int idx, float discount, const float *vp, const float *val, const float *lg,
                                           int horizon, const lz_traverse_args &a, float delta, const int32_t *vtp, hipStream_t s)
{
    if (t.variant == LZ_TREE_EFFICIENTZERO)
        hipLaunchKernelGGL((k_backprop_traverse_wide<LZ_TREE_EFFICIENTZERO>), dim3(t.B), dim3(64), 0, s, t, idx, discount, vp, val, lg, horizon, a, delta, vtp);
    else
        hipLaunchKernelGGL((k_backprop_traverse_wide<LZ_TREE_MUZERO>), dim3(t.B), dim3(64), 0, s, t, idx, discount, vp, val, lg, horizon, a, delta, vtp);
}
