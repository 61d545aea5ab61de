// lz_tree_dev.h -- device-side tree code shared by the tree kernels (lz_tree.hip) and the convolution chain that runs a
// root's tree step as its prologue (lz_nn.hip): per-root view, selection (cbatch_traverse), expansion + backup
// (cbatch_backpropagate), and the step on an LDS copy of the tree.  See lz_tree.hip for the reference citations.
//
// Everything here reproduces the reference's float arithmetic bit for bit, so FMA contraction is switched off for this
// header whatever the including translation unit is built with (restored at the end when the includer asks for it).
#pragma once
#include "lz_internal.h"
#include "lz_wave.h"

#pragma clang fp contract(off)
#include "lz_math.h"  // below the pragma: its polynomial evaluations must stay individually rounded too

#define LZ_FLOAT_MAX 1000000.0f  // cminimax.h:9
#define LZ_FLOAT_MIN (-LZ_FLOAT_MAX)

// bytes of the staged arrays of dev_step_lds for `idx` + 1 nodes, rounded up to 16
__host__ __device__ static inline size_t lz_tree_lds_bytes_raw(int A, int idx) { return (((size_t)(idx + 1) * ((size_t)A * 20 + 28)) + 15) / 16 * 16; }

// lz_tree_dev::node_link: how an expanded node hangs in its tree -- parent node (24 bits) | action at the parent (16 bits) | depth (24 bits;
// root = 0).  Written once when the node is expanded; the tree-parallel selection (dev_traverse_par) reads it.
__host__ __device__ static inline uint64_t lz_link_pack(int parent, int act, int depth)
{
    return ((uint64_t)(uint32_t)parent << 40) | ((uint64_t)((uint32_t)act & 0xffffu) << 24) | (uint64_t)((uint32_t)depth & 0xffffffu);
}

namespace {


__device__ __forceinline__ float rl_f(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ int rl_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uni_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ float mm_normalize(float v, float mn, float mx, float delta_max)
{
    const float d = mx - mn;  // cminimax.cpp:33-45
    if (d > 0) {
        if (d < delta_max) v = (v - mn) / delta_max;
        else v = (v - mn) / d;
    }
    return v;
}

// ------------------------------------------------------------------------------------------------
// Per-root view of the tree arrays.  The kernels below are written once against this view and instantiated
// twice: on the HBM arrays directly, and on an LDS copy of the root's tree (loads from LDS, stores written through to
// HBM) -- a search step is a chain of dependent loads (node -> edges -> child -> ...), ~1 us each from HBM/L2.
// ------------------------------------------------------------------------------------------------
struct tview {
    float4 *edge;            // [NN][A]   read / write here
    int32_t *child;          // [NN][A]
    float *node_vp;          // [NN]
    int32_t *node_reset, *node_to_play;
    const int32_t *path_node, *path_act;  // [NN] path of the previous traverse (read by the backup)
    uint64_t *link;          // [NN] LDS copy of node_link (null on the HBM view: dev_backprop then writes the HBM array only)
    float4 *g_edge;          // write-through targets in HBM (WT instantiation), same indexing
    int32_t *g_child;
    float *g_node_vp;
    int32_t *g_node_reset, *g_node_to_play;
};

// values that cross from the backup into the next selection in registers (fused kernel) or come from HBM
template <int NC>
struct tscal {
    int n_root;        // number of legal root actions
    int root_act[NC];  // this lane's root legal action(s)
    int root_visit;
    float root_vsum;
    float mn, mx;      // CMinMaxStats
    uint32_t epoch;
};

// The store of a timing stamp, kept out of the compiler's sight.  An ordinary global store -- even one under `if (stamp)` that never
// executes in production -- in front of the step's loads makes the compiler treat all global memory as possibly clobbered: every
// wave-uniform load behind it (n_legal[b], rng_epoch[0], vtp[b], ...) is then a vector load + v_readfirstlane with an s_waitcnt vmcnt(0)
// in front, i.e. one DRAINED round trip per uniform value before the tree's own loads are even requested (two of them, ~1 us each, in
// the ISA of the tree-fused chain kernels), instead of an s_load that returns beside them.  Nobody reads a stamp inside the launch.
__device__ __forceinline__ void lz_stamp_store(unsigned long long *p, unsigned long long v)
{
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v));
}

__device__ __forceinline__ tview global_view(const lz_tree_dev &t, int b)
{
    tview v;
    const size_t e = (size_t)b * t.NN * t.A, n = (size_t)b * t.NN;
    v.edge = v.g_edge = t.edge + e;
    v.child = v.g_child = t.child + e;
    v.node_vp = v.g_node_vp = t.node_vp + n;
    v.node_reset = v.g_node_reset = t.node_reset + n;
    v.node_to_play = v.g_node_to_play = t.node_to_play + n;
    v.path_node = t.path_node + n;
    v.path_act = t.path_act + n;
    v.link = nullptr;
    return v;
}

template <int NC>
__device__ __forceinline__ void load_scalars(const lz_tree_dev &t, int b, tscal<NC> &sc)
{
    const int lane = threadIdx.x;
    sc.n_root = uni(t.n_legal[b]);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int j = c * 64 + lane;
        sc.root_act[c] = (j < sc.n_root) ? t.legal[(size_t)b * t.A + j] : 0;
    }
    sc.root_visit = t.root_visit[b];
    sc.root_vsum = t.root_vsum[b];
    sc.mn = t.minmax[2 * b];
    sc.mx = t.minmax[2 * b + 1];
    sc.epoch = t.rng_epoch ? t.rng_epoch[0] : 0u;  // bumped by every prepare: decorrelates env-steps
}

// ------------------------------------------------------------------------------------------------
// traverse: cbatch_traverse (cnode.cpp:886-963) -- select down to an unexpanded child
// ------------------------------------------------------------------------------------------------
// REUSE (ReZero, cnode.cpp:697-754, 816-884, 965-1072): at the root the trajectory's true action is scored by carm_score
// (its value term is the reuse value and, once visited, it gets no prior term), and the walk stops right below the root
// when that action is selected; res_noinf marks roots whose reached node is already expanded (reference index -1).
template <int NC, int VARIANT, bool REUSE = false>
__device__ __forceinline__ void dev_traverse(const lz_tree_dev &t, const tview &v, const tscal<NC> &sc, const lz_traverse_args &a,
                                             float delta_max, int vtp, int true_action = -1, float reuse_value = 0.0f,
                                             int32_t *s_out = nullptr, bool use_tab = false, float tab_pbc = 0.0f, float tab_sq = 0.0f, int b_in = -1)
{
    const int b = b_in >= 0 ? b_in : (int)blockIdx.x, lane = threadIdx.x;   // (b_in: the root of a workgroup whose block id is not its root -- k_sim_fused)
    const int A = t.A, NN = t.NN;
    const float mn = sc.mn, mx = sc.mx;
    const float discount = a.discount;
    const float base = (float)a.pb_c_base;
    const uint32_t epoch = sc.epoch;
    int node = 0, depth = 0, is_root = 1, last_action = -1, noinf = 0;
    int node_visit = sc.root_visit;
    // the path record of level d stays in lane d & 63 and is written once after the walk (coalesced), not by three
    // single-lane stores per level
    int my_node = 0, my_act = 0;
    auto flush_path = [&](int base, int count) {
        if (lane < count) {
            t.path_node[(size_t)b * NN + base + lane] = my_node;
            t.path_act[(size_t)b * NN + base + lane] = my_act;
            t.node_best[(size_t)b * NN + my_node] = my_act;
        }
    };
    float parent_q = 0.0f;

    for (;;) {
        const int n = is_root ? sc.n_root : A;
        const float node_vp = v.node_vp[node];
        const int node_reset = v.node_reset[node];
        float prior[NC], val[NC], tr[NC], score[NC];
        int vis[NC], act[NC], chd[NC];  // chd: child node ids, fetched with the edges (one round trip per level)
        // ---- load the children (one 16-byte edge per lane) and compute_mean_q (cnode.cpp:173-212)
        float total = 0.0f;
        int nv = 0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int j = c * 64 + lane;
            const bool valid = j < n;
            act[c] = valid ? (is_root ? sc.root_act[c] : j) : 0;
            float4 e = valid ? v.edge[(size_t)node * A + act[c]] : make_float4(0.f, 0.f, 0.f, 0.f);
            chd[c] = valid ? v.child[(size_t)node * A + act[c]] : -1;
            prior[c] = e.x;
            vis[c] = __float_as_int(e.y);
            val[c] = (vis[c] == 0) ? 0.0f : e.z / (float)vis[c];  // CNode::value cnode.cpp:223-239
            if (VARIANT == LZ_TREE_EFFICIENTZERO) {
                tr[c] = e.w - node_vp;
                if (node_reset == 1) tr[c] = e.w;
            } else {
                tr[c] = e.w;
            }
            const float qsa = tr[c] + discount * val[c];
            uint64_t mask = __ballot(valid && vis[c] > 0);
            while (mask) {  // total_unsigned_q += qsa in legal-list order
                const int j2 = __builtin_ctzll(mask);
                total += rl_f(qsa, j2);
                nv += 1;
                mask &= mask - 1;
            }
        }
        float mean_q;
        if (is_root && nv > 0) mean_q = total / (float)nv;
        else mean_q = (parent_q + total) / (float)(nv + 1);
        const int was_root = is_root;
        is_root = 0;
        parent_q = mean_q;

        // ---- cucb_score (cnode.cpp:756-814) for every child
        // use_tab: lane n holds both factors for N = n (computed once per launch, while the tree is being staged): the
        // software logf with its table fetch and the sqrt leave the per-level dependent chain.  Same expressions, same bits.
        const float N = (float)(node_visit - 1);
        float pbc0, sq;
        if (use_tab) {
            pbc0 = rl_f(tab_pbc, node_visit - 1);
            sq = rl_f(tab_sq, node_visit - 1);
        } else {
            pbc0 = lz_logf((N + base + 1) / base) + a.pb_c_init;
            sq = sqrtf(N);
        }
        float best = -__builtin_inff();
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int j = c * 64 + lane;
            float pb_c = pbc0 * (sq / (float)(vis[c] + 1));
            const float prior_score = pb_c * prior[c];
            const bool arm = REUSE && was_root && act[c] == true_action;  // carm_score instead of cucb_score
            const float vchild = arm ? reuse_value : val[c];
            float value_score;
            if (vis[c] == 0) value_score = mean_q;
            else if (a.players == 1) value_score = tr[c] + discount * vchild;
            else value_score = tr[c] + discount * (-vchild);
            value_score = mm_normalize(value_score, mn, mx, delta_max);
            if (value_score < 0) value_score = 0;
            else if (value_score > 1) value_score = 1;
            float ucb = prior_score + value_score;
            if (arm && vis[c] != 0) ucb = value_score;
            score[c] = (j < n) ? ucb : -__builtin_inff();
            best = fmaxf(best, score[c]);
        }
        best = (NC == 1 && n <= 16) ? row0_max(best) : wave_max(best);
        // ---- cselect_child (cnode.cpp:651-695): front of the tie list == first arg-max in list order
        int pos = -1;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const uint64_t mask = __ballot(score[c] == best);
            if (pos < 0 && mask) pos = c * 64 + __builtin_ctzll(mask);
        }
        if (a.tiebreak == LZ_TIE_RANDOM && pos >= 0) {
            // tie list = [first arg-max] + later entries with score >= max - 1e-6 (cnode.cpp:675-685)
            const float thr = best - 0.000001f;
            uint64_t masks[NC];
            int cnt = 0;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int j = c * 64 + lane;
                masks[c] = __ballot(j == pos || (j > pos && score[c] >= thr));
                cnt += __builtin_popcountll(masks[c]);
            }
            if (cnt > 1) {  // a single candidate (the usual case once visits differ) is the first arg-max: `pos` stays, no draw, no scan
                const uint64_t h = mix64(mix64(a.seed ^ ((uint64_t)epoch << 20) ^ (uint64_t)a.counter) ^ ((uint64_t)b << 12) ^ (uint64_t)depth);
                int r = (int)(((h >> 32) * (uint64_t)cnt) >> 32);  // uniform index in [0, cnt) without a 64-bit division
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    uint64_t mk = masks[c];
                    const int pc = __builtin_popcountll(mk);
                    if (r >= 0 && r < pc) {
                        for (int q = 0; q < r; ++q) mk &= mk - 1;
                        pos = c * 64 + __builtin_ctzll(mk);
                        r = -1;
                    } else if (r >= pc) {
                        r -= pc;
                    }
                }
            }
        }
        int action = 0, sel_visit = 0, nxt = -3;
        if (pos >= 0 && best > LZ_FLOAT_MIN) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if ((pos >> 6) == c) {
                    action = rl_i(act[c], pos & 63);
                    sel_visit = rl_i(vis[c], pos & 63);
                    nxt = rl_i(chd[c], pos & 63);
                }
            }
        }
        if (nxt == -3) {
            // no child entered the reference's tie list (every score NaN: a node whose logits all lie below FLOAT_MIN = -1e6 has
            // priors 0 / 0, cnode.cpp:123-137): `action` stays 0 (cnode.cpp:687-693).  The walk continues below that child
            // with ITS visit count like any other step.
            nxt = uni(v.child[(size_t)node * A + action]);
            sel_visit = uni(__float_as_int(v.edge[(size_t)node * A + action].y));
        }
        if (a.players > 1) vtp = (vtp == 1) ? 2 : 1;  // cnode.cpp:932-943
        if (lane == (depth & 63)) { my_node = node; my_act = action; }
        last_action = action;
        depth += 1;
        if ((depth & 63) == 0) flush_path(depth - 64, 64);
        if (REUSE && was_root && action == true_action) { noinf = nxt >= 0 ? 1 : 0; break; }  // cnode.cpp:1041-1044
        if (nxt < 0) break;  // reached an unexpanded child: the leaf
        node = nxt;
        node_visit = sel_visit;
    }
    flush_path((depth - 1) & ~63, depth - ((depth - 1) & ~63));
    if (lane == 0) {
        t.res_ix[b] = node;  // parent->current_latent_state_index (cnode.cpp:955); stays a valid slot when noinf
        t.res_iy[b] = REUSE ? (noinf ? b : t.node_bidx[(size_t)b * NN + node]) : b;  // parent->batch_index
        t.res_last_action[b] = last_action;
        t.res_search_len[b] = depth;
        t.res_vtp[b] = vtp;
        if (REUSE) t.res_noinf[b] = noinf;
        if (s_out) { s_out[0] = node; s_out[1] = last_action; }
    }
}

// ------------------------------------------------------------------------------------------------
// traverse, tree-parallel form (trees of at most 64 expanded nodes with at most AU actions, on the LDS copy): the same
// cbatch_traverse (cnode.cpp:886-963), but EVERY expanded node picks its child at once -- lane n owns node n -- and the walk from
// the root then only follows the choices.  dev_traverse scores one node per level with 6 of 64 lanes busy (A = 6) and ~600
// dependent instructions per level, so a search path of depth d costs ~2.4 k cycles x d; here the cost is one pass over the
// tree whatever the depth:
//   A  lane n reads its node's A edges and sums compute_mean_q's total / count IN THE LANE, in legal-list order
//      (cnode.cpp:173-212: the same additions in the same order as the v_readlane replay of dev_traverse);
//   B  parent_q chains down the tree: mean_q[n] = (mean_q[parent] + total[n]) / (count[n] + 1), one ds_bpermute + one
//      division per LEVEL of the tree for all nodes of that level (node_link carries parent and depth);
//   C  cucb_score (cnode.cpp:756-814) of every child and cselect_child's arg-max / tie list (cnode.cpp:651-695) in the lane,
//      children in list order; the tie-break draw is keyed by the node's depth exactly like the level counter of the walk;
//   D  the walk: two v_readlane per level.
// Every float operation of a node is the one dev_traverse performs for that node, in the same order => bit-identical selections
// (tests/test_tree_gpu.py / test_exact_replay_gpu.py run both; LZ_TRAVERSE_SERIAL=1 selects the per-level walk).
// s_tab: [64] log((n + base + 1) / base) + init, then [64] sqrt(n) -- the exploration factors of a node with visit count n + 1.
template <int AU, int VARIANT>
__device__ __forceinline__ void dev_traverse_par(const lz_tree_dev &t, const tview &v, const tscal<1> &sc, const lz_traverse_args &a,
                                                 float delta_max, int vtp, int nn, const float *s_tab, int32_t *s_out, int b_in = -1)
{
    const int b = b_in >= 0 ? b_in : (int)blockIdx.x, lane = threadIdx.x;
    const int A = t.A, NN = t.NN;
    const float mn = sc.mn, mx = sc.mx, discount = a.discount;
    const bool nvalid = lane < nn;
    const int n = nvalid ? lane : nn - 1;           // (idle lanes recompute the last node: every address stays inside the staged tree)
    const bool is_root = n == 0;
    // ---- A: this node's record, its link, its children.  Below the root child j is action j: the A records are consecutive
    // (immediate offsets; for A < AU the reads run on into the next node's records -- or, behind the last node, into the arrays
    // that follow the edges in the staged tree -- and everything beyond the list is masked below).  The root's children are its legal
    // list: when that is the identity (every Atari root) the root is a node like any other, else lane 0 takes its addresses from
    // the list.
    const uint64_t link = v.link[n];
    const int parent = (int)(link >> 40), act_in = (int)((link >> 24) & 0xffffu), depth_n = (int)(link & 0xffffffu);
    const float node_vp = v.node_vp[n];
    const int node_reset = v.node_reset[n];
    const int cnt_n = is_root ? sc.n_root : A;      // children in the list: the legal root actions | every action
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f e[AU];
    int chd[AU];
    const bool ident = sc.n_root == A && __ballot(lane < A && sc.root_act[0] != lane) == 0;   // wave-uniform
    if (ident) {
#pragma unroll
        for (int j = 0; j < AU; ++j) {
            e[j] = *reinterpret_cast<const v4f *>(v.edge + (size_t)n * A + j);
            chd[j] = v.child[(size_t)n * A + j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < AU; ++j) {
            const int ra = rl_i(sc.root_act[0], j);  // legal[j] of the root (0 beyond the list)
            const int aj = is_root ? ra : j;
            e[j] = *reinterpret_cast<const v4f *>(v.edge + (size_t)n * A + aj);
            chd[j] = v.child[(size_t)n * A + aj];
        }
    }
    const int chd_a0 = v.child[(size_t)n * A];      // child of action 0: where a node without any comparable score goes (cnode.cpp:687-693)
    // visit count of the edge that leads here (the walk's node_visit); the root carries its own
    const int in_vis_e = __float_as_int(v.edge[(size_t)parent * A + act_in].y);
    const int in_vis = is_root ? sc.root_visit : in_vis_e;
    float prior[AU], val[AU], tr[AU];
    int vis[AU];
    float total = 0.0f;
    int nv = 0;
#pragma unroll
    for (int j = 0; j < AU; ++j) {
        prior[j] = e[j].x;
        vis[j] = __float_as_int(e[j].y);
        // CNode::value cnode.cpp:223-239.  (Branch-free on purpose: the division runs on a harmless denominator where the
        // count is 0 and a select drops it -- one instruction stream for all lanes instead of a divergent branch per child.)
        const float qv = e[j].z / (float)max(vis[j], 1);
        val[j] = (vis[j] == 0) ? 0.0f : qv;
        if (VARIANT == LZ_TREE_EFFICIENTZERO) {
            const float dv = e[j].w - node_vp;
            tr[j] = (node_reset == 1) ? e[j].w : dv;
        } else {
            tr[j] = e[j].w;
        }
        const float qsa = tr[j] + discount * val[j];
        const bool visited = j < cnt_n && vis[j] > 0;   // total_unsigned_q += qsa in legal-list order
        const float t2 = total + qsa;
        total = visited ? t2 : total;
        nv += visited ? 1 : 0;
    }
    // ---- B: mean_q of every node, level by level (the root's parent_q is 0)
    const float rden = (float)(nv + 1);
    float mq = (is_root && nv > 0) ? total / (float)nv : (0.0f + total) / rden;
    for (int lvl = 1; __ballot(nvalid && depth_n >= lvl) != 0; ++lvl) {
        const float pq = __int_as_float(__builtin_amdgcn_ds_bpermute(parent << 2, __float_as_int(mq)));
        const float m2 = (pq + total) / rden;
        mq = (depth_n == lvl) ? m2 : mq;
    }
    // ---- C: cucb_score of every child, cselect_child
    const int ti = min(max(in_vis - 1, 0), 63);
    const float pbc0 = s_tab[ti], sq = s_tab[64 + ti];
    float score[AU];
    float best = -__builtin_inff();
    const float mm_d = mx - mn;
    const bool mm_on = mm_d > 0;
    const float mm_den = (mm_d < delta_max) ? delta_max : mm_d;
#pragma unroll
    for (int j = 0; j < AU; ++j) {
        float pb_c = pbc0 * (sq / (float)(vis[j] + 1));
        const float prior_score = pb_c * prior[j];
        const float vq = (a.players == 1) ? tr[j] + discount * val[j] : tr[j] + discount * (-val[j]);
        float value_score = (vis[j] == 0) ? mq : vq;
        // CMinMaxStats::normalize (cminimax.cpp:33-45), branch-free: mm_den is the denominator the reference divides by when
        // maximum > minimum (wave-uniform), the quotient is dropped otherwise
        const float nq = (value_score - mn) / mm_den;
        value_score = mm_on ? nq : value_score;
        value_score = (value_score < 0) ? 0.0f : ((value_score > 1) ? 1.0f : value_score);
        const float ucb = prior_score + value_score;
        score[j] = (j < cnt_n) ? ucb : -__builtin_inff();
        best = fmaxf(best, score[j]);
    }
    int pos = -1;   // front of the tie list == first arg-max in list order
#pragma unroll
    for (int j = AU - 1; j >= 0; --j) pos = (score[j] == best) ? j : pos;
    const bool ok = pos >= 0 && best > LZ_FLOAT_MIN;
    if (a.tiebreak == LZ_TIE_RANDOM) {
        // tie list = [first arg-max] + later entries with score >= max - 1e-6 (cnode.cpp:675-685)
        const float thr = best - 0.000001f;
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < AU; ++j) cnt += (j == pos || (j > pos && score[j] >= thr)) ? 1 : 0;
        const bool draw = ok && cnt > 1;
        if (__ballot(draw)) {   // (a single candidate -- the usual case once visits differ -- needs no draw)
            const uint64_t h = mix64(mix64(a.seed ^ ((uint64_t)sc.epoch << 20) ^ (uint64_t)a.counter) ^ ((uint64_t)b << 12) ^ (uint64_t)depth_n);
            int r = (int)(((h >> 32) * (uint64_t)cnt) >> 32);  // uniform index in [0, cnt)
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
    int sel_child = chd_a0;
#pragma unroll
    for (int j = 0; j < AU; ++j) sel_child = (ok && pos == j) ? chd[j] : sel_child;
    int sel_act = ok ? pos : 0;   // below the root (and at a root with the identity list) the action IS the list position
    if (!ident) {                 // the root's action comes from its legal list
        const int p0 = rl_i(sel_act, 0), ok0 = (int)(__ballot(ok) & 1ull);
        const int a0 = ok0 ? rl_i(sc.root_act[0], p0) : 0;
        sel_act = (lane == 0) ? a0 : sel_act;
    }
    // ---- D: the walk
    int node = 0, depth = 0, last_action = -1;
    int my_node = 0, my_act = 0;
    for (;;) {
        const int action = rl_i(sel_act, node), nxt = rl_i(sel_child, node);
        if (a.players > 1) vtp = (vtp == 1) ? 2 : 1;  // cnode.cpp:932-943
        if (lane == depth) { my_node = node; my_act = action; }
        last_action = action;
        depth += 1;
        if (nxt < 0 || depth >= 64) break;  // reached an unexpanded child: the leaf  (a path has at most nn <= 64 nodes)
        node = nxt;
    }
    if (lane < depth) {
        t.path_node[(size_t)b * NN + lane] = my_node;
        t.path_act[(size_t)b * NN + lane] = my_act;
        t.node_best[(size_t)b * NN + my_node] = my_act;
    }
    if (lane == 0) {
        t.res_ix[b] = node;
        t.res_iy[b] = b;
        t.res_last_action[b] = last_action;
        t.res_search_len[b] = depth;
        t.res_vtp[b] = vtp;
        if (s_out) { s_out[0] = node; s_out[1] = last_action; }
    }
}

// ------------------------------------------------------------------------------------------------
// backpropagate: cbatch_backpropagate (cnode.cpp:577-601) = expand the leaf, then cbackpropagate.
// d = search length of the path, lg[] = this lane's policy logits of the leaf, sc carries root visit / value sum and
// the min-max statistics in and out.
// ------------------------------------------------------------------------------------------------
// no_expand (ReZero, cnode.cpp:626-630): the leaf is an already expanded node -- nothing is expanded, its own value prefix
// stays, only is_reset is refreshed and value_b (the reuse value) is backed up.  bidx = the leaf's batch_index.
// The new node's priors: softmax of the leaf's policy logits in the reference's order of operations (cnode.cpp:88-151: maximum, exp of the
// differences by the libm-exact lz_expf, the sum over the actions in index order, one division per action).  Lane j of chunk c gets action
// c * 64 + j's.  A function of the policy logits only.
template <int NC>
__device__ __forceinline__ void dev_expand_priors(const float (&lg)[NC], int A, const uint64_t *exptab, float (&pri)[NC])
{
    float e[NC];
    float m = LZ_FLOAT_MIN;
#pragma unroll
    for (int c = 0; c < NC; ++c) m = fmaxf(m, lg[c]);
    m = wave_max(m);
#pragma unroll
    for (int c = 0; c < NC; ++c) e[c] = lz_expf_core(lg[c] - m, exptab);
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int cnt = min(64, A - c * 64);
        for (int j = 0; j < cnt; ++j) sum += rl_f(e[c], j);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) pri[c] = e[c] / sum;
}

template <int NC, int VARIANT, bool WT>
__device__ __forceinline__ void dev_backprop(const lz_tree_dev &t, const tview &v, tscal<NC> &sc, int new_node, float discount,
                                             float vp_b, float value_b, const float (&lg)[NC], int d, int to_play, int reset,
                                             bool no_expand = false, int bidx = -1, const uint64_t *exptab = nullptr,
                                             const float *prior_in = nullptr, int b_in = -1)
{
    const int b = b_in >= 0 ? b_in : (int)blockIdx.x, lane = threadIdx.x;
    const int A = t.A, NN = t.NN;
    // ---- CNode::expand (cnode.cpp:88-151): all A actions are legal below the root
    if (!no_expand) {
        float pri[NC];
        if (prior_in) {   // dev_expand_priors ran already (dev_step_lds with split heads: on the policy logits, before value / value prefix were out)
#pragma unroll
            for (int c = 0; c < NC; ++c) pri[c] = prior_in[c];
        } else {
            dev_expand_priors<NC>(lg, A, exptab, pri);
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int j = c * 64 + lane;
            if (j < A) {
                const size_t o = (size_t)new_node * A + j;
                const float4 ne = make_float4(pri[c], __int_as_float(0), 0.0f, 0.0f);
                v.edge[o] = ne;
                v.child[o] = -1;
                if (WT) { v.g_edge[o] = ne; v.g_child[o] = -1; }
            }
        }
    }
    const int parent = uni(v.path_node[d - 1]);
    const int pact = uni(v.path_act[d - 1]);
    if (no_expand) {
        if (VARIANT == LZ_TREE_EFFICIENTZERO && lane == 0) {
            const int leaf = v.child[(size_t)parent * A + pact];
            v.node_reset[leaf] = reset;
            if (WT) v.g_node_reset[leaf] = reset;
        }
    } else if (lane == 0) {
        t.node_bidx[(size_t)b * NN + new_node] = bidx < 0 ? b : bidx;
        const uint64_t lk = lz_link_pack(parent, pact, d);
        t.node_link[(size_t)b * NN + new_node] = lk;
        if (v.link) v.link[new_node] = lk;
        v.child[(size_t)parent * A + pact] = new_node;
        v.node_vp[new_node] = vp_b;
        v.node_reset[new_node] = reset;
        v.node_to_play[new_node] = to_play;
        if (WT) {
            v.g_child[(size_t)parent * A + pact] = new_node;
            v.g_node_vp[new_node] = vp_b;
            v.g_node_reset[new_node] = reset;
            v.g_node_to_play[new_node] = to_play;
        }
        t.node_best[(size_t)b * NN + new_node] = -1;
    }
    if (WT) __builtin_amdgcn_wave_barrier();
    // ---- cbackpropagate (cnode.cpp:482-575): path node P_k, k = d (leaf) .. 0 (root); lane i of a
    // chunk owns k = d - (chunk*64 + i).  Gather is parallel, the bootstrap recurrence is a scalar chain.
    float bootstrap = value_b;
    float mn = sc.mn, mx = sc.mx;
    for (int k0 = d; k0 >= 0; k0 -= 64) {
        const int k = k0 - lane;
        const bool valid = k >= 0;
        int pn = 0, pa = 0, vis = 0, own_tp = to_play, parent_reset = 0;
        float prior = 0.f, vsum = 0.f, own_vp = 0.f, parent_vp = 0.f;
        if (valid) {
            if (k >= 1) {
                pn = v.path_node[k - 1];
                pa = v.path_act[k - 1];
                const float4 e = v.edge[(size_t)pn * A + pa];
                prior = e.x;
                vis = __float_as_int(e.y);
                vsum = e.z;
                own_vp = (k == d && !no_expand) ? vp_b : e.w;
                parent_vp = v.node_vp[pn];
                parent_reset = v.node_reset[pn];
                if (k < d) own_tp = v.node_to_play[v.path_node[k]];
            } else {
                vis = sc.root_visit;
                vsum = sc.root_vsum;
                own_vp = v.node_vp[0];
                own_tp = v.node_to_play[0];
            }
        }
        float true_reward, tr_eff;
        if (VARIANT == LZ_TREE_EFFICIENTZERO) {
            true_reward = own_vp - parent_vp;
            tr_eff = (parent_reset == 1) ? own_vp : true_reward;
        } else {
            true_reward = own_vp;
            tr_eff = own_vp;
        }
        const int same = (to_play == -1) ? 1 : (own_tp == to_play ? 1 : 0);
        const int cnt = min(64, k0 + 1);
        float my_boot = 0.0f;
        for (int i = 0; i < cnt; ++i) {
            if (lane == i) my_boot = bootstrap;
            const float tre = rl_f(tr_eff, i);
            if (to_play == -1) bootstrap = tre + discount * bootstrap;
            else if (rl_i(same, i)) bootstrap = -tre + discount * bootstrap;
            else bootstrap = tre + discount * bootstrap;
        }
        int new_root_visit = 0;
        float new_root_vsum = 0.0f;
        float q = 0.0f;
        if (valid) {
            vsum = same ? vsum + my_boot : vsum + (-my_boot);
            vis += 1;
            const float value = vsum / (float)vis;
            if (VARIANT == LZ_TREE_EFFICIENTZERO) q = true_reward + discount * value;  // cnode.cpp:516/:558
            else q = (to_play == -1) ? true_reward + discount * value : true_reward + discount * -value;
            if (k >= 1) {
                const float4 ne = make_float4(prior, __int_as_float(vis), vsum, own_vp);
                v.edge[(size_t)pn * A + pa] = ne;
                if (WT) v.g_edge[(size_t)pn * A + pa] = ne;
            } else {
                t.root_visit[b] = vis;
                t.root_vsum[b] = vsum;
                new_root_visit = vis;
                new_root_vsum = vsum;
            }
        }
        minmax_update_ordered(q, valid, mn, mx);
        // the root is lane k0 of the chunk that contains k == 0
        if (k0 < 64) {
            sc.root_visit = rl_i(new_root_visit, k0);
            sc.root_vsum = rl_f(new_root_vsum, k0);
        }
    }
    if (lane == 0) { t.minmax[2 * b] = mn; t.minmax[2 * b + 1] = mx; }
    sc.mn = mn;
    sc.mx = mx;
}

// leaf inputs of the backup: search length, player, value prefix, value, logits, is_reset
template <int NC, int VARIANT>
struct leaf_in {
    int d, to_play, reset;
    float vp, value;
    float lg[NC];
};
template <int NC, int VARIANT>
__device__ __forceinline__ void load_leaf(const lz_tree_dev &t, int b, const float *__restrict__ vps, const float *__restrict__ values,
                                          const float *__restrict__ logits, const int32_t *__restrict__ is_reset, int horizon,
                                          const int32_t *__restrict__ to_play_in, leaf_in<NC, VARIANT> &L)
{
    const int lane = threadIdx.x, A = t.A;
    L.d = uni(t.res_search_len[b]);
    L.to_play = uni(to_play_in ? to_play_in[b] : t.res_vtp[b]);
    L.vp = vps[b];
    L.value = values[b];
    L.reset = 0;
    if (VARIANT == LZ_TREE_EFFICIENTZERO) {
        if (is_reset) L.reset = is_reset[b];
        else if (horizon > 0) L.reset = (L.d % horizon == 0) ? 1 : 0;  // mcts_ctree.py:859
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int j = c * 64 + lane;
        L.lg[c] = (j < A) ? logits[(size_t)b * A + j] : LZ_FLOAT_MIN;
    }
}

// expand + backup of the previous simulation followed by the selection of the next one, on an LDS copy of the root's tree:
// every array the step reads is requested in the first instructions (one HBM round trip for the whole tree: edges, child
// ids, node records, the previous path, the leaf's network outputs and the root scalars), the backup and the selection
// then chase pointers inside LDS, and every store is written through to HBM.  One wavefront; s_tree needs
// lz_tree_lds_bytes(t, new_node) bytes.  s_out (optional, LDS): [0] = selected parent slot (res_ix), [1] = last action.
// s_leaf (optional, LDS): the leaf's network outputs arrive HERE instead of in vps / values / logits -- [0] value prefix, [1] value,
// [2 .. 2 + A) policy logits, written by other waves of the workgroup, which then raise s_leaf_flag[0] to leaf_ready (split heads,
// k_chain_w): the tree is staged first, then this wave waits for the flag.
template <int NC, int VARIANT>
__device__ __forceinline__ void dev_step_lds(const lz_tree_dev &t, int b, int new_node, float discount,
                                             const float *__restrict__ vps, const float *__restrict__ values,
                                             const float *__restrict__ logits, int horizon, const lz_traverse_args &a,
                                             float delta_max, const int32_t *__restrict__ vtp_in, float4 *s_tree,
                                             int32_t *s_out = nullptr, unsigned long long *ts = nullptr,
                                             const float *s_leaf = nullptr, const int32_t *s_leaf_flag = nullptr, int leaf_ready = 0)
{
    const int lane = threadIdx.x;
    const bool stamp = ts && b == 0 && lane == 0;   // timing experiments only (ts is null in production)
#define LZ_TTS(i) do { if (stamp) lz_stamp_store(ts + (i), __builtin_readcyclecounter()); } while (0)
    LZ_TTS(0);
    const int A = t.A;
    const int nn = new_node + 1;        // nodes 0 .. new_node exist after this step
    float4 *s_edge = s_tree;                                              // [nn][A]
    uint64_t *s_link = reinterpret_cast<uint64_t *>(s_edge + (size_t)nn * A);  // [nn] node_link (parent | action | depth)
    int32_t *s_child = reinterpret_cast<int32_t *>(s_link + nn);         // [nn][A]
    float *s_vp = reinterpret_cast<float *>(s_child + (size_t)nn * A);    // [nn]
    int32_t *s_reset = reinterpret_cast<int32_t *>(s_vp + nn);
    int32_t *s_tp = s_reset + nn;
    int32_t *s_pn = s_tp + nn;
    int32_t *s_pa = s_pn + nn;
    uint64_t *s_exp = reinterpret_cast<uint64_t *>(s_tree + lz_tree_lds_bytes_raw(A, new_node) / 16);  // 32 entries, 16-byte aligned
    float *s_tab = reinterpret_cast<float *>(s_exp + 32);                 // [2][64] exploration factors by visit count (dev_traverse_par)
    const tview g = global_view(t, b);
    const uint64_t *g_link = t.node_link + (size_t)b * t.NN;
    // ---- one round trip: everything the step reads.  All requests are unconditional (clamped indices) and issued before
    // the first use: a predicated load, or a wave-uniform read of a loaded value in between, would split this into
    // several dependent HBM/L2 round trips (~0.6 us each).
    // The per-root scalars are wave-uniform loads.  Left to the compiler each becomes a vector load + v_readfirstlane placed right behind it,
    // with an s_waitcnt vmcnt(0) in between (s_load is not selected: the barriers / fences in front may have clobbered memory as far as it
    // knows) -- a drained round trip per value before the tree's loads are requested (ISA of the tree-fused chain kernels: two of them).  An
    // index the compiler cannot see through (zero, from an asm) keeps them ordinary per-lane loads of this one batch; they are made uniform
    // where they are used, after the staging.  For the same reason no request is conditional: null pointers are replaced by valid memory.
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    const int bz = b + z;
    const int r_nroot = t.n_legal[bz], r_visit = t.root_visit[bz], r_d = t.res_search_len[bz], r_tp = t.res_vtp[bz];
    const float r_vsum = t.root_vsum[bz], r_mn = t.minmax[2 * bz], r_mx = t.minmax[2 * bz + 1];
    const float *vps_p = s_leaf ? t.root_vsum : vps, *val_p = s_leaf ? t.root_vsum : values;
    const float *lg_p = s_leaf ? reinterpret_cast<const float *>(t.legal) : logits;
    float r_vp = vps_p[bz], r_val = val_p[bz];
    const uint32_t *ep_p = t.rng_epoch ? t.rng_epoch : reinterpret_cast<const uint32_t *>(t.n_legal);
    const uint32_t r_epoch_raw = ep_p[z];
    const uint32_t r_epoch = t.rng_epoch ? r_epoch_raw : 0u;
    const int vtp_raw = vtp_in[bz];
    int r_act[NC];
    float r_lg[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int j = min(c * 64 + lane, A - 1);
        r_act[c] = t.legal[(size_t)b * A + j];
        r_lg[c] = lg_p[(size_t)b * A + j];
    }
    const int ne = new_node * A;        // edges / child ids of the existing nodes 0 .. new_node - 1 (ne >= 1)
    // first batch of every array: all requests, then the table arithmetic (it runs while they are in flight), then the LDS stores;
    // two separate load-then-store loops would be two dependent round trips
    constexpr int UE = 8, UN = 2;
    typedef float v4f __attribute__((ext_vector_type(4)));  // native vectors: arrays of HIP's float4 struct end up in scratch here
    v4f e0[UE];
    int32_t ch0[UE];
#pragma unroll
    for (int u = 0; u < UE; ++u) {
        const int i = min(u * 64 + lane, ne - 1);
        e0[u] = *reinterpret_cast<const v4f *>(g.edge + i);
        ch0[u] = g.child[i];
    }
    float nv0[UN];
    int32_t nr0[UN], nt0[UN], pn0[UN], pa0[UN];
    uint64_t lk0[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        const int i = min(u * 64 + lane, new_node - 1);
        nv0[u] = g.node_vp[i]; nr0[u] = g.node_reset[i]; nt0[u] = g.node_to_play[i]; pn0[u] = g.path_node[i]; pa0[u] = g.path_act[i];
        lk0[u] = g_link[i];
    }
    // tables that take software exp / log table fetches and the sqrt out of the dependent chains below: the 2^(i/32) table of
    // lz_expf goes to LDS, and lane n computes the exploration factors of a node with visit count n + 1 (N = n <= new_node)
    const uint64_t etab = lz_exp2f_tab((unsigned)(lane & 31));   // (a fetch from the constant segment: requested with the rest, stored below)
    const bool use_tab = new_node < 64;
    const float *tabp = a.tab ? a.tab : t.root_vsum;              // per-search table (lz_tree_launch_explore_tab); unconditional requests again
    float tab_pbc = tabp[a.tab ? lane : 0], tab_sq = tabp[a.tab ? 64 + lane : 0];
    if (!a.tab) {
        const float nf = (float)lane;
        tab_pbc = lz_logf((nf + (float)a.pb_c_base + 1) / (float)a.pb_c_base) + a.pb_c_init;
        tab_sq = sqrtf(nf);
    }
    __builtin_amdgcn_sched_barrier(0);
    LZ_TTS(1);
    if (lane < 32) s_exp[lane] = etab;
#pragma unroll
    for (int u = 0; u < UE; ++u) {
        const int i = u * 64 + lane;
        if (i < ne) { *reinterpret_cast<v4f *>(s_edge + i) = e0[u]; s_child[i] = ch0[u]; }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        const int i = u * 64 + lane;
        if (i < new_node) { s_vp[i] = nv0[u]; s_reset[i] = nr0[u]; s_tp[i] = nt0[u]; s_pn[i] = pn0[u]; s_pa[i] = pa0[u]; s_link[i] = lk0[u]; }
    }
    s_tab[lane] = tab_pbc;
    s_tab[64 + lane] = tab_sq;
    for (int i0 = UE * 64; i0 < ne; i0 += UE * 64) {  // trees beyond 512 edges / 128 nodes: further batches
        v4f e[UE];
        int32_t ch[UE];
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            const int i = min(i0 + u * 64 + lane, ne - 1);
            e[u] = *reinterpret_cast<const v4f *>(g.edge + i);
            ch[u] = g.child[i];
        }
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            const int i = i0 + u * 64 + lane;
            if (i < ne) { *reinterpret_cast<v4f *>(s_edge + i) = e[u]; s_child[i] = ch[u]; }
        }
    }
    for (int i0 = UN * 64; i0 < new_node; i0 += UN * 64) {
        float nv[UN];
        int32_t nr[UN], nt[UN], pn[UN], pa[UN];
        uint64_t lk[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = min(i0 + u * 64 + lane, new_node - 1);
            nv[u] = g.node_vp[i]; nr[u] = g.node_reset[i]; nt[u] = g.node_to_play[i]; pn[u] = g.path_node[i]; pa[u] = g.path_act[i];
            lk[u] = g_link[i];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = i0 + u * 64 + lane;
            if (i < new_node) { s_vp[i] = nv[u]; s_reset[i] = nr[u]; s_tp[i] = nt[u]; s_pn[i] = pn[u]; s_pa[i] = pa[u]; s_link[i] = lk[u]; }
        }
    }
    float pri_early[NC];
    if (s_leaf) {
        // The tree is staged; now the leaf's network outputs, once the waves that compute them say so.  The policy head is the first to
        // finish (one wave, A outputs; the value and value-prefix heads are three waves of 601 outputs each and the second of them gets its
        // weights 2.4 k cycles after the first: a CU's 64 B/clk): the new node's priors -- the software exp, the ordered sum and the
        // divisions of the expansion -- are computed on the logits alone (s_leaf_flag[1]) while those two finish.
        while (__hip_atomic_load(s_leaf_flag + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
        float lg_m[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            r_lg[c] = s_leaf[2 + min(c * 64 + lane, A - 1)];
            lg_m[c] = (c * 64 + lane < A) ? r_lg[c] : LZ_FLOAT_MIN;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // (the exp table in LDS was written by this wave's lanes above)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        dev_expand_priors<NC>(lg_m, A, s_exp, pri_early);
        while (__hip_atomic_load(s_leaf_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != leaf_ready) __builtin_amdgcn_s_sleep(1);
        r_vp = s_leaf[0];
        r_val = s_leaf[1];
    }
    // the same values load_scalars / load_leaf produce
    tscal<NC> sc;
    sc.n_root = uni(r_nroot);
    sc.root_visit = r_visit; sc.root_vsum = r_vsum; sc.mn = r_mn; sc.mx = r_mx; sc.epoch = r_epoch;
    leaf_in<NC, VARIANT> L;
    L.d = uni(r_d);
    L.to_play = uni(r_tp);
    const int vtp = uni(vtp_raw);
    L.vp = r_vp; L.value = r_val;
    L.reset = (VARIANT == LZ_TREE_EFFICIENTZERO && horizon > 0) ? ((L.d % horizon == 0) ? 1 : 0) : 0;  // mcts_ctree.py:859
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int j = c * 64 + lane;
        sc.root_act[c] = (j < sc.n_root) ? r_act[c] : 0;
        L.lg[c] = (j < A) ? r_lg[c] : LZ_FLOAT_MIN;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    LZ_TTS(2);
    tview v = g;
    v.edge = s_edge; v.child = s_child; v.node_vp = s_vp; v.node_reset = s_reset; v.node_to_play = s_tp;
    v.path_node = s_pn; v.path_act = s_pa; v.link = s_link;
    dev_backprop<NC, VARIANT, true>(t, v, sc, new_node, discount, L.vp, L.value, L.lg, L.d, L.to_play, L.reset, false, -1, s_exp, s_leaf ? pri_early : nullptr, b);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    LZ_TTS(3);
    // trees of at most 64 nodes and 8 actions (every Atari-sized search of up to 63 simulations): all nodes scored at once
    if constexpr (NC == 1) {
        if (use_tab && A <= 8 && !a.serial) {
            if (A <= 4) dev_traverse_par<4, VARIANT>(t, v, sc, a, delta_max, vtp, nn, s_tab, s_out, b);
            else if (A <= 6) dev_traverse_par<6, VARIANT>(t, v, sc, a, delta_max, vtp, nn, s_tab, s_out, b);
            else dev_traverse_par<8, VARIANT>(t, v, sc, a, delta_max, vtp, nn, s_tab, s_out, b);
        }
        else dev_traverse<NC, VARIANT>(t, v, sc, a, delta_max, vtp, -1, 0.0f, s_out, use_tab, tab_pbc, tab_sq, b);
    } else {
        dev_traverse<NC, VARIANT>(t, v, sc, a, delta_max, vtp, -1, 0.0f, s_out, use_tab, tab_pbc, tab_sq, b);
    }
    LZ_TTS(4);
#undef LZ_TTS
}

}  // namespace

// LDS bytes of the staged tree of one root after `idx` nodes exist besides the new one (dev_step_lds): the arrays, rounded up to
// 16 bytes, then the 32-entry exp table and the two 64-entry exploration-factor tables
static inline size_t lz_tree_lds_bytes(const lz_tree_dev &t, int idx) { return lz_tree_lds_bytes_raw(t.A, idx) + 32 * 8 + 128 * 4; }

// Largest staged tree (bytes per root) the LDS step is used for; beyond it the step walks the HBM arrays.  Staging costs
// O(tree) per simulation but every level of the walk then costs an LDS instead of an HBM round trip.
// LZ_TREE_LDS_LIMIT overrides (experiments).
static inline size_t lz_tree_lds_limit(size_t dflt)
{
    const char *v = getenv("LZ_TREE_LDS_LIMIT");
    return v && *v ? (size_t)strtoul(v, nullptr, 0) : dflt;
}

#ifdef LZ_TREE_DEV_RESTORE_FAST_CONTRACT  // defined by includers that are NOT built with -ffp-contract=off (lz_nn.hip)
#pragma clang fp contract(fast)          // hipcc's default for device code
#endif
