// lz_tree_sampled.hip -- on-device Sampled-EfficientZero tree (continuous action spaces) for gfx950.
//
// Same machinery as lz_tree.hip (one wavefront per root, edge records in HBM, reference-order float sums, exact
// libm-compatible logf, -ffp-contract=off), with the sampled-action twists of
//   lzero/mcts/ctree/ctree_sampled_efficientzero/lib/cnode.cpp
//       expand :194-282,:408-452   K children per node, actions = tanh(N(mu, sigma)), keyed by to_string hashes
//       compute_mean_q :480-520, cbackpropagate :860-945, cselect_child :968-1024, cucb_score :1026-1108 (the
//       shipped "uniform" prior branch pb_c * 1 / children.size()), cbatch_traverse :1110-1187.
// A node's K legal-action positions map onto child slots through rep[] (position of the first action with the
// same "%f" rendering in every dimension): duplicated actions share one child exactly like the reference's
// std::map<size_t, CNode>, while every position still takes part in the select / mean-Q / distribution loops.
//
// Draws: `given` (caller-supplied, bit-exact parity against the oracle's libstdc++ restatement) or on-device
// (counter-based hash -> Box-Muller -> mu + sigma z -> tanhf), which matches the reference in distribution only
// -- the reference itself seeds a std::default_random_engine from the wall clock in every expand.
#include "lz_internal.h"
#include "lz_math.h"
#include "lz_wave.h"

#define LZ_FLOAT_MAX 1000000.0f
#define LZ_FLOAT_MIN (-LZ_FLOAT_MAX)

namespace {

__device__ __forceinline__ float rl_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ int rl_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float mm_normalize(float v, float mn, float mx, float delta_max)
{
    const float d = mx - mn;
    if (d > 0) {
        if (d < delta_max) v = (v - mn) / delta_max;
        else v = (v - mn) / d;
    }
    return v;
}

// std::to_string(float) == "%f": two values print identically iff they round to the same 6-decimal number with the
// same sign ("-0.000000" differs from "0.000000"); compared through the bits of rint(x * 1e6).
__device__ __forceinline__ uint64_t fkey(float x) { return (uint64_t)__double_as_longlong(rint((double)x * 1000000.0)); }

// draws (or copies) the K x D actions of node `node` of root b into t.actions and LDS, fills rep[] / nchild and
// initialises the K edge records.  One wavefront; lane i owns legal position i.
__device__ __forceinline__ void expand_sampled(const lz_tree_dev &t, int b, int node, const lz_sample_args &sa,
                                               float *s_act /* LDS [K][D] */)
{
    const int lane = threadIdx.x, K = t.A, D = t.D, NN = t.NN;
    float *gact = t.actions + (((size_t)b * NN + node) * K) * D;
    if (t.disc_A > 0 && !sa.given) {
        // discrete action space (cnode.cpp:288-327): K of the A actions without replacement -- the reference sorts the keys
        // u_a^(1/p_a) in descending order and keeps the first K; log(u_a) / p_a orders the same way.  Lane l owns actions l, l + 64,
        // l + 128, l + 192 (A <= 256: bipedalwalker_cont_disc_sampled_efficientzero_config.py's 4^4, mujoco_disc's 5^3); an action's
        // rank is counted against every key by v_readlane, chunk by chunk.  With A <= 64 only chunk 0 exists: the draw of rounds 4-5.
        constexpr int NCD = 4;
        const int A = t.disc_A;
        const uint32_t epoch = t.rng_epoch ? t.rng_epoch[0] : 0u;
        float ex[NCD], key[NCD];
        float sum = 0.0f;
#pragma unroll
        for (int c = 0; c < NCD; ++c) {
            const int a = c * 64 + lane;
            const float lg = a < A ? sa.policy[(size_t)b * A + a] : -__builtin_inff();
            ex[c] = a < A ? expf(lg) : 0.0f;
            if (c == 0) sum = ex[0];
            else if (c * 64 < A) sum += ex[c];
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
#pragma unroll
        for (int c = 0; c < NCD; ++c) {
            const int a = c * 64 + lane;
            const float p = ex[c] / (sum + 1e-6f);
            const uint64_t st = mix64(mix64(sa.seed ^ 0x5a3c1e0fu ^ ((uint64_t)epoch << 24) ^ (uint64_t)sa.counter) ^ ((uint64_t)b << 20) ^
                                      ((uint64_t)node << 8) ^ (uint64_t)a);
            const float u = ((float)((st >> 40) & 0xffffff) + 0.5f) * (1.0f / 16777216.0f);
            key[c] = a < A ? logf(u) / fmaxf(p, 1e-30f) : -__builtin_inff();
        }
        int rank[NCD] = {0, 0, 0, 0};
#pragma unroll
        for (int c2 = 0; c2 < NCD; ++c2) {
            const int cnt = min(64, A - c2 * 64);   // <= 0 beyond the last chunk
            for (int l2 = 0; l2 < cnt; ++l2) {
                const float k2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(key[c2]), l2));
                const int a2 = c2 * 64 + l2;
#pragma unroll
                for (int c = 0; c < NCD; ++c) rank[c] += (k2 > key[c] || (k2 == key[c] && a2 < c * 64 + lane)) ? 1 : 0;
            }
        }
#pragma unroll
        for (int c = 0; c < NCD; ++c)
            if (c * 64 + lane < A && rank[c] < K) s_act[rank[c]] = (float)(c * 64 + lane);
        __syncthreads();
        if (lane < K) gact[lane] = s_act[lane];
    } else if (lane < K) {
        if (sa.given) {
            for (int j = 0; j < D; ++j) s_act[lane * D + j] = sa.given[((size_t)b * K + lane) * D + j];
        } else {
            const uint32_t epoch = t.rng_epoch ? t.rng_epoch[0] : 0u;
            uint64_t st = mix64(mix64(sa.seed ^ 0x5a3c1e0fu ^ ((uint64_t)epoch << 24) ^ (uint64_t)sa.counter) ^ ((uint64_t)b << 20) ^
                                ((uint64_t)node << 8) ^ (uint64_t)lane);
            for (int j = 0; j < D; ++j) {
                st = mix64(st);
                const float u1 = ((float)((st >> 40) & 0xffffff) + 0.5f) * (1.0f / 16777216.0f);
                const float u2 = ((float)((st >> 16) & 0xffffff) + 0.5f) * (1.0f / 16777216.0f);
                // hardware log / cos / exp / rcp (~1e-6): these draws are compared with nothing bit for bit (the reference seeds
                // its generator from the clock; parity runs inject the draws)
                const float z = __fsqrt_rn(-2.0f * __logf(u1)) * __cosf(6.28318530718f * u2);
                const float mu = sa.policy[(size_t)b * 2 * D + j], sigma = sa.policy[(size_t)b * 2 * D + D + j];
                s_act[lane * D + j] = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * (mu + sigma * z)));
            }
        }
        for (int j = 0; j < D; ++j) gact[lane * D + j] = s_act[lane * D + j];
    }
    __syncthreads();
    // rep = the first position whose D keys all equal this lane's (a key class is represented by its first member).  The keys
    // travel by v_readlane: per dimension one fkey per lane and K uniform steps, instead of a per-lane double loop over LDS
    // with two fkeys per comparison.
    uint64_t eq = ~0ull;  // bit i2: position i2 matches this lane on every dimension so far
    for (int j = 0; j < D; ++j) {
        const uint64_t kj = fkey(s_act[min(lane, K - 1) * D + j]);
        const int klo = (int)(uint32_t)kj, khi = (int)(uint32_t)(kj >> 32);
        uint64_t m = 0;
        for (int i2 = 0; i2 < K; ++i2) {
            const int lo2 = __builtin_amdgcn_readlane(klo, i2), hi2 = __builtin_amdgcn_readlane(khi, i2);
            m |= (uint64_t)((lo2 == klo) & (hi2 == khi)) << i2;
        }
        eq &= m;
    }
    const int rp = lane < K ? __builtin_ctzll(eq | (1ull << lane)) : lane;  // eq always contains the lane itself
    const uint64_t firsts = __ballot(lane < K && rp == lane);
    if (lane < K) {
        const size_t o = ((size_t)b * NN + node) * K + lane;
        t.rep[o] = rp;
        t.edge[o] = make_float4(0.0f, __int_as_float(0), 0.0f, 0.0f);
        t.child[o] = -1;
    }
    if (lane == 0) t.nchild[(size_t)b * NN + node] = __builtin_popcountll(firsts);
}

__global__ __launch_bounds__(64) void k_sprepare(lz_tree_dev t, lz_sample_args sa, const float *__restrict__ vps,
                                                 const int32_t *__restrict__ to_play)
{
    extern __shared__ float s_act[];
    const int b = blockIdx.x, lane = threadIdx.x;
    expand_sampled(t, b, 0, sa, s_act);
    if (lane == 0) {
        const size_t o = (size_t)b * t.NN;
        t.node_vp[o] = vps[b];
        t.node_reset[o] = 0;
        t.node_to_play[o] = to_play[b];
        t.node_best[o] = -1;
        t.root_visit[b] = 1;
        t.root_vsum[b] = 0.0f;
    }
}

__device__ __forceinline__ void dev_straverse(const lz_tree_dev &t, const lz_traverse_args &a, float delta_max,
                                              const int32_t *__restrict__ vtp_in)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int K = t.A, NN = t.NN, D = t.D;
    const float4 *edge_b = t.edge + (size_t)b * NN * K;
    const int32_t *child_b = t.child + (size_t)b * NN * K;
    const int32_t *rep_b = t.rep + (size_t)b * NN * K;
    const float mn = t.minmax[2 * b], mx = t.minmax[2 * b + 1];
    const float discount = a.discount, base = (float)a.pb_c_base;
    int vtp = vtp_in[b];
    const uint32_t epoch = t.rng_epoch ? t.rng_epoch[0] : 0u;
    int node = 0, depth = 0, is_root = 1, last_pos = 0;
    int node_visit = t.root_visit[b];
    float parent_q = 0.0f;
    for (;;) {
        const bool valid = lane < K;
        const float node_vp = t.node_vp[(size_t)b * NN + node];
        const int node_reset = t.node_reset[(size_t)b * NN + node];
        const int nch = t.nchild[(size_t)b * NN + node];
        // one round trip per level: every lane fetches its OWN slot (unconditional, clamped) next to rep[], then takes the
        // representative's record from lane rep through the LDS crossbar -- edge[rep[lane]] as a dependent load was a second one
        const int lc = min(lane, K - 1);
        const int rp_raw = rep_b[(size_t)node * K + lc];
        const float4 e_own = edge_b[(size_t)node * K + lc];
        const int chd_own = child_b[(size_t)node * K + lc];
        const int rp = valid ? rp_raw : 0;
        float4 e;
        e.x = __shfl(e_own.x, rp); e.y = __shfl(e_own.y, rp); e.z = __shfl(e_own.z, rp); e.w = __shfl(e_own.w, rp);
        int chd = __shfl(chd_own, rp);
        if (!valid) { e = make_float4(0.f, 0.f, 0.f, 0.f); chd = -1; }
        const int vis = __float_as_int(e.y);
        const float val = (vis == 0) ? 0.0f : e.z / (float)vis;
        float tr = e.w - node_vp;
        if (node_reset == 1) tr = e.w;
        // compute_mean_q (cnode.cpp:480-520): every legal position, duplicates included, in order
        const float qsa = tr + discount * val;
        float total = 0.0f;
        int nv = 0;
        uint64_t mask = __ballot(valid && vis > 0);
        while (mask) {
            const int j2 = __builtin_ctzll(mask);
            total += rl_f(qsa, j2);
            nv += 1;
            mask &= mask - 1;
        }
        float mean_q;
        if (is_root && nv > 0) mean_q = total / (float)nv;
        else mean_q = (parent_q + total) / (float)(nv + 1);
        is_root = 0;
        parent_q = mean_q;
        // cucb_score (cnode.cpp:1026-1108), uniform prior branch
        const float N = (float)(node_visit - 1);
        const float pbc0 = lz_logf((N + base + 1) / base) + a.pb_c_init;
        const float sq = sqrtf(N);
        const float pb_c = pbc0 * (sq / (float)(vis + 1));
        const float prior_score = pb_c * 1 / (float)nch;
        float value_score;
        if (vis == 0) value_score = mean_q;
        else if (a.players == 1) value_score = tr + discount * val;
        else value_score = tr + discount * (-val);
        value_score = mm_normalize(value_score, mn, mx, delta_max);
        if (value_score < 0) value_score = 0;
        if (value_score > 1) value_score = 1;
        const float score = valid ? prior_score + value_score : -__builtin_inff();
        const float best = wave_max(score);
        int pos = __builtin_ctzll(__ballot(score == best));
        if (a.tiebreak == LZ_TIE_RANDOM) {
            const float thr = best - 0.000001f;
            uint64_t mk = __ballot(lane == pos || (lane > pos && score >= thr));
            const int cnt = __builtin_popcountll(mk);
            if (cnt > 1) {  // a single candidate (the usual case once visits differ) needs no draw
                const uint64_t h = mix64(mix64(a.seed ^ ((uint64_t)epoch << 20) ^ (uint64_t)a.counter) ^ ((uint64_t)b << 12) ^ (uint64_t)depth);
                const int r = (int)(((h >> 32) * (uint64_t)cnt) >> 32);  // uniform index in [0, cnt) without a 64-bit division
                for (int q = 0; q < r; ++q) mk &= mk - 1;
                pos = __builtin_ctzll(mk);
            }
        }
        const int rpos = rl_i(rp, pos);
        const int nxt = rl_i(chd, pos);
        const int sel_visit = rl_i(vis, pos);
        if (a.players > 1) vtp = (vtp == 1) ? 2 : 1;
        if (lane == 0) {
            t.node_best[(size_t)b * NN + node] = pos;
            t.path_node[(size_t)b * NN + depth] = node;
            t.path_act[(size_t)b * NN + depth] = rpos;  // the slot that carries the child's statistics
        }
        last_pos = pos;
        depth += 1;
        if (nxt < 0) break;
        node = nxt;
        node_visit = sel_visit;
    }
    if (lane < D) t.res_last_action_f[(size_t)b * D + lane] = t.actions[(((size_t)b * NN + node) * K + last_pos) * D + lane];
    for (int j = 64 + lane; j < D; j += 64) t.res_last_action_f[(size_t)b * D + j] = t.actions[(((size_t)b * NN + node) * K + last_pos) * D + j];
    if (lane == 0) {
        t.res_ix[b] = node;
        t.res_iy[b] = b;
        // discrete action spaces: the network wants the action index (one-hot encoding), not its position among the K samples
        t.res_last_action[b] = t.disc_A > 0 ? (int)t.actions[(((size_t)b * NN + node) * K + last_pos) * D] : last_pos;
        t.res_search_len[b] = depth;
        t.res_vtp[b] = vtp;
    }
}

__device__ __forceinline__ void dev_sbackprop(const lz_tree_dev &t, int new_node, float discount,
                                              const float *__restrict__ vps, const float *__restrict__ values,
                                              const lz_sample_args &sa, const int32_t *__restrict__ is_reset, int horizon,
                                              const int32_t *__restrict__ to_play_in, float *s_act)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int K = t.A, NN = t.NN;
    float4 *edge_b = t.edge + (size_t)b * NN * K;
    int32_t *child_b = t.child + (size_t)b * NN * K;
    // Everything that does not depend on another load is requested here, unconditionally, before the first wave-uniform read:
    // leaf scalars, statistics, the root record and the WHOLE previous path (position = lane).  The backup then needs one
    // more round trip (the edge / node records of the path entries) instead of five dependent ones.
    const size_t nb = (size_t)b * NN;
    const int d_raw = t.res_search_len[b];
    const int tp_raw = to_play_in ? to_play_in[b] : t.res_vtp[b];
    const float vp_b = vps[b];
    const float value_b = values[b];
    const int rst_raw = is_reset ? is_reset[b] : 0;
    float mn = t.minmax[2 * b], mx = t.minmax[2 * b + 1];
    const int root_visit = t.root_visit[b];
    const float root_vsum = t.root_vsum[b], root_vp = t.node_vp[nb];
    const int root_tp = t.node_to_play[nb];
    const int pl = min(lane, NN - 1);
    const int pnode_l = t.path_node[nb + pl], pact_l = t.path_act[nb + pl];
    expand_sampled(t, b, new_node, sa, s_act);
    const int d = uni(d_raw);
    const int to_play = uni(tp_raw);
    int reset = 0;
    if (is_reset) reset = rst_raw;
    else if (horizon > 0) reset = (d % horizon == 0) ? 1 : 0;
    const bool short_path = d <= 64;  // every path position sits in a lane (always, unless a search is deeper than 64)
    const int parent = short_path ? rl_i(pnode_l, d - 1) : uni(t.path_node[nb + d - 1]);
    const int pact = short_path ? rl_i(pact_l, d - 1) : uni(t.path_act[nb + d - 1]);
    if (lane == 0) {
        child_b[(size_t)parent * K + pact] = new_node;
        const size_t o = nb + new_node;
        t.node_vp[o] = vp_b;
        t.node_reset[o] = reset;
        t.node_to_play[o] = to_play;
        t.node_best[o] = -1;
    }
    // cbackpropagate (cnode.cpp:860-945): identical to the EfficientZero tree
    float bootstrap = value_b;
    for (int k0 = d; k0 >= 0; k0 -= 64) {
        const int k = k0 - lane;
        const bool valid = k >= 0;
        int pn = 0, pa = 0, vis = 0, own_tp = to_play, parent_reset = 0;
        float prior = 0.f, vsum = 0.f, own_vp = 0.f, parent_vp = 0.f;
        if (short_path) {
            // path entries by lane exchange, then ONE batch of unconditional loads (clamped to entry 0 where k < 1)
            const bool inner = valid && k >= 1;
            pn = __shfl(pnode_l, max(k - 1, 0));
            pa = __shfl(pact_l, max(k - 1, 0));
            const int pnk = __shfl(pnode_l, min(max(k, 0), 63));
            pn = inner ? pn : 0;
            pa = inner ? pa : 0;
            const float4 e = edge_b[(size_t)pn * K + pa];
            const float pvp = t.node_vp[nb + pn];
            const int prs = t.node_reset[nb + pn];
            const int otp = t.node_to_play[nb + ((inner && k < d) ? pnk : 0)];
            if (inner) {
                prior = e.x;
                vis = __float_as_int(e.y);
                vsum = e.z;
                own_vp = (k == d) ? vp_b : e.w;
                parent_vp = pvp;
                parent_reset = prs;
                if (k < d) own_tp = otp;
            } else if (valid) {
                vis = root_visit;
                vsum = root_vsum;
                own_vp = root_vp;
                own_tp = root_tp;
            }
        } else if (valid) {
            if (k >= 1) {
                pn = t.path_node[nb + k - 1];
                pa = t.path_act[nb + k - 1];
                const float4 e = edge_b[(size_t)pn * K + pa];
                prior = e.x;
                vis = __float_as_int(e.y);
                vsum = e.z;
                own_vp = (k == d) ? vp_b : e.w;
                parent_vp = t.node_vp[nb + pn];
                parent_reset = t.node_reset[nb + pn];
                if (k < d) own_tp = t.node_to_play[nb + t.path_node[nb + k]];
            } else {
                vis = root_visit;
                vsum = root_vsum;
                own_vp = root_vp;
                own_tp = root_tp;
            }
        }
        const float true_reward = own_vp - parent_vp;
        const float tr_eff = (parent_reset == 1) ? own_vp : true_reward;
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
        float q = 0.0f;
        if (valid) {
            vsum = same ? vsum + my_boot : vsum + (-my_boot);
            vis += 1;
            const float value = vsum / (float)vis;
            q = true_reward + discount * value;
            if (k >= 1) edge_b[(size_t)pn * K + pa] = make_float4(prior, __int_as_float(vis), vsum, own_vp);
            else { t.root_visit[b] = vis; t.root_vsum[b] = vsum; }
        }
        minmax_update_ordered(q, valid, mn, mx);
    }
    if (lane == 0) { t.minmax[2 * b] = mn; t.minmax[2 * b + 1] = mx; }
}

__global__ __launch_bounds__(64) void k_straverse(lz_tree_dev t, lz_traverse_args a, float delta_max,
                                                  const int32_t *__restrict__ vtp_in)
{
    dev_straverse(t, a, delta_max, vtp_in);
}

__global__ __launch_bounds__(64) void k_sbackprop(lz_tree_dev t, int new_node, float discount,
                                                  const float *__restrict__ vps, const float *__restrict__ values,
                                                  lz_sample_args sa, const int32_t *__restrict__ is_reset, int horizon,
                                                  const int32_t *__restrict__ to_play_in)
{
    extern __shared__ float s_act[];
    dev_sbackprop(t, new_node, discount, vps, values, sa, is_reset, horizon, to_play_in, s_act);
}

// expand + backup of simulation s followed by the selection of simulation s + 1 for the same root, in one launch
__global__ __launch_bounds__(64) void k_sbackprop_straverse(lz_tree_dev t, int new_node, float discount,
                                                            const float *__restrict__ vps, const float *__restrict__ values,
                                                            lz_sample_args sa, int horizon, lz_traverse_args a, float delta_max,
                                                            const int32_t *__restrict__ vtp_in)
{
    extern __shared__ float s_act[];
    dev_sbackprop(t, new_node, discount, vps, values, sa, nullptr, horizon, nullptr, s_act);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    dev_straverse(t, a, delta_max, vtp_in);
}

// get_children_distribution (cnode.cpp:570-590): the visit count of every legal position's (possibly shared) child
__global__ void k_sreadout(lz_tree_dev t, int32_t *__restrict__ dist, float *__restrict__ values)
{
    const int b = blockIdx.x, K = t.A, NN = t.NN;
    for (int j = threadIdx.x; j < K; j += blockDim.x)
        dist[(size_t)b * K + j] = __float_as_int(t.edge[((size_t)b * NN) * K + t.rep[((size_t)b * NN) * K + j]].y);
    if (threadIdx.x == 0 && values) {
        const int rv = t.root_visit[b];
        values[b] = (rv == 0) ? 0.0f : t.root_vsum[b] / (float)rv;
    }
}

}  // namespace

void lz_stree_launch_prepare(const lz_tree_dev &t, const lz_sample_args &sa, const float *d_vp, const int32_t *d_to_play, hipStream_t s)
{
    hipLaunchKernelGGL(k_sprepare, dim3(t.B), dim3(64), (size_t)t.A * t.D * 4, s, t, sa, d_vp, d_to_play);
}
void lz_stree_launch_traverse(const lz_tree_dev &t, const lz_traverse_args &a, float delta, const int32_t *d_vtp_in, hipStream_t s)
{
    hipLaunchKernelGGL(k_straverse, dim3(t.B), dim3(64), 0, s, t, a, delta, d_vtp_in);
}
void lz_stree_launch_backprop(const lz_tree_dev &t, int latent_index, float discount, const float *d_vp, const float *d_values,
                              const lz_sample_args &sa, const int32_t *d_is_reset, int horizon, const int32_t *d_to_play, hipStream_t s)
{
    hipLaunchKernelGGL(k_sbackprop, dim3(t.B), dim3(64), (size_t)t.A * t.D * 4, s, t, latent_index, discount, d_vp, d_values, sa,
                       d_is_reset, horizon, d_to_play);
}
void lz_stree_launch_backprop_traverse(const lz_tree_dev &t, int latent_index, float discount, const float *d_vp, const float *d_values,
                                       const lz_sample_args &sa, int horizon, const lz_traverse_args &a, float delta,
                                       const int32_t *d_vtp_in, hipStream_t s)
{
    hipLaunchKernelGGL(k_sbackprop_straverse, dim3(t.B), dim3(64), (size_t)t.A * t.D * 4, s, t, latent_index, discount, d_vp, d_values, sa,
                       horizon, a, delta, d_vtp_in);
}
void lz_stree_launch_readout(const lz_tree_dev &t, int32_t *d_dist, float *d_values, hipStream_t s)
{
    hipLaunchKernelGGL(k_sreadout, dim3(t.B), dim3(64), 0, s, t, d_dist, d_values);
}
