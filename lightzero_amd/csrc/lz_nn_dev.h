// lz_nn_dev.h -- device helpers shared by the network translation units (lz_nn.hip, lz_chain_s3g.hip): vector typedefs, the activation
// functions, the exact three-term bf16 split, global-address-space weight pointers, write-through stores, in-graph stamps.
// Include AFTER lz_tree_dev.h (lz_stamp_store, lz_tree_step).
#pragma once
#include <hip/hip_runtime.h>

#include "lz_hinv.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4;

// ---- the resident search (LZ_RESIDENT=1; k_search_resident in lz_nn.hip): the launch-per-simulation kernels' bodies run inside ONE launch,
// once per simulation.  lz_res_sim says which simulation of the launch a body call is: every per-simulation pointer of the launch
// sequence (leaf slot, its output rows, the next latent's pool slot, the LSTM's output slot) is linear in it.
struct lz_res_sim {
    int ds;                 // simulations since the launch's first one
    int B, BA;              // roots; roots x actions (policy logits per pool slot)
    long long lat_step;     // floats between two latent pool slots
    long long hc_step;      // floats between two h / c pool slots (B x H)
};

namespace {

// Loads of data ANOTHER workgroup of the same launch stored (plain stores, then s_waitcnt vmcnt(0), then a flag): sc1 -- the load bypasses
// this CU's vector L1 (which other CUs' stores never refresh) and is served by the XCD's L2; through a buffer descriptor / a relaxed
// agent-scope atomic so that the compiler tracks its completion like any other load (MI355X_MICROARCH.md, inter-workgroup visibility).
// Correct only for producers on the SAME XCD (the resident search's groups of 16 roots are formed by HW_REG_XCC_ID).
typedef unsigned lz_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 load_sc1_f4(const float *base, size_t float_off)
{
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0x7ffffff0, 0x00020000);
    const lz_v4u u = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(float_off * 4), 0, 16);
    return __builtin_bit_cast(f32x4, u);
}
__device__ __forceinline__ float load_sc1_f(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int load_sc1_i(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int VEC> struct vecf;
template <> struct vecf<4> { typedef float4 type; };
template <> struct vecf<2> { typedef float2 type; };

__device__ __forceinline__ float vget(const float4 &v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }
__device__ __forceinline__ float vget(const float2 &v, int j) { return j == 0 ? v.x : v.y; }
__device__ __forceinline__ float4 vzero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// The activation of a network: ReLU (every shipped EfficientZero / MuZero conv configuration) or GELU(approximate='tanh') -- the default of the
// convolutional Sampled EfficientZero (sampled_efficientzero_model.py:40), which its Atari configuration keeps.  GELU instances are separate
// template instantiations (bool GELU): the ReLU kernels' code does not change.  tanh(y) = 1 - 2 / (1 + e^{2y}) on the hardware exp / rcp as in
// lz_dense.hip (|error| < 3e-7 absolute).
__device__ __forceinline__ float gelu_tanh_(float u)
{
    const float y = 0.7978845608028654f * (u + 0.044715f * u * u * u);
    const float t = 1.0f - 2.0f * __frcp_rn(1.0f + __expf(2.0f * y));
    return 0.5f * u * (1.0f + t);
}
template <bool GELU> __device__ __forceinline__ float act_(float v) { if constexpr (GELU) return gelu_tanh_(v); else return fmaxf(v, 0.0f); }

struct no_step {};
template <int TREE> struct step_arg { typedef lz_tree_step type; };
template <> struct step_arg<0> { typedef no_step type; };

// A pointer rebuilt from an integer (v_readlane of a per-layer address kept in a lane) is a FLAT pointer to the compiler: its loads become
// flat_load, which count on vmcnt AND lgkmcnt and may return out of order with LDS traffic, so every wait on them is s_waitcnt vmcnt(0)
// lgkmcnt(0) -- a full drain of the weight stream in the middle of the MFMA loop (seen in the ISA of k_chain_s3 and k_chain_b).  Say that
// the address is global.
typedef __attribute__((address_space(1))) const bf16x8 gbl_bf16x8;
__device__ __forceinline__ gbl_bf16x8 *as_global_bf16x8(unsigned long long addr) { return (gbl_bf16x8 *)addr; }
__device__ __forceinline__ bf16x8 gload(gbl_bf16x8 *p) { return *p; }

// Write-through stores (sc0 sc1) for the big per-launch outputs of the recurrent loop (next latent, head-convolution rows, LSTM state, head
// partials: 4-5 MB per launch).  What a kernel leaves dirty in L2 is written back at the kernel boundary, in front of the next launch: measured
// (fast mode, same box, in-graph stamps) the gap behind the chain launch 3.4 -> 2.9 us and behind the LSTM launch 2.5 -> 1.95 us with these
// stores written through while the kernel still runs.  Nobody reads them back inside the launch.
__device__ __forceinline__ void store_wt(float *p, const f32x4 &v)
{
    // The s_nop belongs to the store: a VMEM store of more than 64 bits reads its data registers late, and a VALU write to them within two wait
    // states corrupts the stored value (gfx940 hazard).  The compiler's hazard recognizer covers its own stores but cannot see into inline
    // assembly -- found when an experiment's register allocation reused the data registers as the next store's address (6 % of the rows wrong).
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_wt(float *p, float v)
{
    asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// in-graph timing (bench.py): the FIRST workgroup of a launch stores the time it starts, the LAST one (by block id) the time it ends
// (s_memrealtime: 100 MHz, independent of the shader clock); st == null in production (one wave-uniform branch).  Plain stores from two
// workgroups: a first version that folded every workgroup's times in by atomics cost the 512-workgroup LSTM launch 3 us.
__device__ __forceinline__ void lz_stamp_begin(unsigned long long *st)
{
    if (st && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) lz_stamp_store(st, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}
__device__ __forceinline__ void lz_stamp_end(unsigned long long *st)
{
    if (st && threadIdx.x == 0 && blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1) lz_stamp_store(st + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}

// hi = rne_bf16(x), mid = rne_bf16(x - hi), lo = rne_bf16(x - hi - mid): hi + mid + lo == x exactly (3 x 8 significant bits >= 24; both
// subtractions are exact), and (hi + mid) + lo evaluated in fp32 returns x bit for bit (tests/test_split_bf16_cpu.py)
__device__ __forceinline__ void split3_bf16(const f32x4 &v, bf16x4 &h, bf16x4 &m, bf16x4 &l)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const __bf16 hq = (__bf16)v[q];
        const float r1 = v[q] - (float)hq;
        const __bf16 mq = (__bf16)r1;
        const float r2 = r1 - (float)mq;
        h[q] = hq; m[q] = mq; l[q] = (__bf16)r2;
    }
}

// ---- split heads: the head MLPs of the PREVIOUS simulation's leaf, finished by waves 1..7 of the root's workgroup while wave 0 stages
// the root's tree (lz_split_heads).  hw = wave - 1: 0..2 value head, 3..5 value-prefix head (601 outputs over 3 waves x 64 lanes x <= 4),
// 6 policy head.  Every wave first sums the first-layer partial blocks of the LSTM launch for its head (32 unit tiles, fixed order: two
// halves of 16 sequentially, then half 0 + half 1), applies bias / BatchNorm / ReLU -> hidden unit j in lanes j and j + 32 -- and
// has requested its second-layer weights right behind them.  The three waves of a categorical head meet ONCE in LDS (each sums
// exp(logit - its own maximum); the first of them waits on a counter for the other two and rescales to the common maximum; wave 0 of
// the workgroup is not part of this); the scalars go to the pool slot and to s_leaf, then s_ctr[2] counts the finished heads (3 = the
// leaf is ready).  s_ctr[0..3] must be zero when this starts.  Measured (tools/tree_timing.py, root 0, simulation 49): the 155 KB of
// second-layer weights of a root pass the CU's vector-memory path (64 B/clk) in ~4.4 k cycles, the scalars are out at ~8-10 k -- the
// tree wave has staged its tree by ~7 k, so ~2-3 k cycles of this remain exposed in the launch.
// RES (resident search): the partials were stored by other workgroups of THIS launch -> sc1 loads; off_b / off_ba: the leaf's slot of this
// simulation relative to the launch's first (floats into out_value | out_vp and out_logits)
template <bool RES = false>
__device__ __forceinline__ void heads_in_prologue(const lz_split_heads &sh, int b, int A, int hw, int lane, float *s_leaf, int32_t *s_ctr,
                                                  float *s_red, unsigned long long *ts = nullptr, size_t off_b = 0, size_t off_ba = 0)
{
    const bool stamp = ts && b == 0 && hw == 0 && lane == 0;   // timing experiments (debug build): stamps of head wave 1 of root 0
#define LZ_HPS(i) do { if (stamp) lz_stamp_store(ts + 8 + (i), __builtin_readcyclecounter()); } while (0)
    LZ_HPS(0);
    const int head = hw < 3 ? 0 : (hw < 6 ? 2 : 1);   // 0 value, 1 policy, 2 value prefix (the order of lz_split_heads' arrays)
    const int gw = hw < 6 ? hw % 3 : 0, grp = hw < 3 ? 0 : 1;
    const int NOUT = head == 1 ? A : sh.nout;
    // ---- requests in the order of use (a wave's loads return in order): the first-layer partials of this root and head -- [32 unit
    // tiles][32 hidden] contiguous; lane (ug = lane >> 3, jq = lane & 7) takes the hidden quad 4 jq .. + 3 of unit tiles ug, ug + 8,
    // ug + 16, ug + 24 --, the first layer's bias / BatchNorm, then the second-layer weights of this lane's outputs
    const int NU = sh.n_unit_tiles, ug = lane >> 3, jq = lane & 7;
    const size_t ppo = ((size_t)b * 3 + head) * (NU * 32) + ug * 32 + jq * 4;
    const float *pp = sh.part + ppo;
    f32x4 pv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if constexpr (RES) pv[q] = load_sc1_f4(sh.part, ppo + (size_t)q * 8 * 32);
        else pv[q] = *reinterpret_cast<const f32x4 *>(pp + (size_t)q * 8 * 32);
    }
    const f32x4 b1v = *reinterpret_cast<const f32x4 *>(sh.b1[head] + jq * 4), s1v = *reinterpret_cast<const f32x4 *>(sh.s1[head] + jq * 4),
                t1v = *reinterpret_cast<const f32x4 *>(sh.t1[head] + jq * 4);
    constexpr int NT = 4;
    f32x4 w2[NT][8];
    float lg[NT];
    const int nstep = head == 1 ? 64 : 192, n0 = gw * 64 + lane;
#pragma unroll
    for (int t = 0; t < NT; ++t) {   // unconditional, clamped: predicating the unused columns away (25 % of the requests) split the burst
        const int n = min(n0 + nstep * t, NOUT - 1);   // into dependent pieces and was measured slower (heads out at 11.5 k instead of 8.2 k cycles)
        lg[t] = sh.b2[head][n];
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) w2[t][k4] = *reinterpret_cast<const f32x4 *>(sh.w2t[head] + ((size_t)k4 * NOUT + n) * 4);
    }
    LZ_HPS(1);
    // ---- hidden units: the four partials of a lane in order, then the eight ug groups over lanes ^ 8, ^ 16, ^ 32 (a fixed order,
    // the same in every lane of a column); bias, BatchNorm, ReLU; one copy per wave in LDS for the broadcast reads below
    f32x4 hid4;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
        float v = ((pv[0][c4] + pv[1][c4]) + pv[2][c4]) + pv[3][c4];
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8 = lane ^ 8
        v = xor32_sum(xor16_sum(v));
        hid4[c4] = fmaxf((v + b1v[c4]) * s1v[c4] + t1v[c4], 0.0f);
    }
    float *s_hid = s_red + 64 + hw * 32;
    if (lane < 8) *reinterpret_cast<f32x4 *>(s_hid + lane * 4) = hid4;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    LZ_HPS(2);
    // ---- second layer: the hidden units come back as broadcast reads (same address in every lane)
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) {
        const f32x4 h4 = *reinterpret_cast<const f32x4 *>(s_hid + k4 * 4);
#pragma unroll
        for (int t = 0; t < NT; ++t)
            lg[t] += (w2[t][k4][0] * h4[0] + w2[t][k4][1] * h4[1]) + (w2[t][k4][2] * h4[2] + w2[t][k4][3] * h4[3]);
    }
    LZ_HPS(3);
    if (sh.dbg_logits && head != 1) {   // parity tests (tracing): the support-wide logits of this simulation's value / value-prefix head
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = n0 + nstep * t;
            if (n < NOUT) sh.dbg_logits[((size_t)grp * sh.dbg_B + b) * NOUT + n] = lg[t];
        }
    }
    if (head == 1) {   // policy logits
        if (lane < A) {
            sh.out_logits[off_ba + (size_t)b * A + lane] = lg[0];
            s_leaf[2 + lane] = lg[0];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) {   // s_ctr[3]: the policy logits alone are out (the tree wave starts the new node's priors on them), s_ctr[2]: one more head done
            __hip_atomic_fetch_add(s_ctr + 3, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(s_ctr + 2, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }
    // ---- softmax . support -> inverse scalar transform.  Each of the head's three waves sums exp(logit - ITS OWN maximum); the three
    // (maximum, sum, weighted sum) triples meet once in LDS and are rescaled to the common maximum: one rendezvous instead of two
    float m = -__builtin_inff();
#pragma unroll
    for (int t = 0; t < NT; ++t) m = (n0 + nstep * t < NOUT) ? fmaxf(m, lg[t]) : m;
    m = wave_max(m);
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n0 + nstep * t;
        if (n < NOUT) {
            const float ex = expf(lg[t] - m);
            s0 += ex;
            s1 += ex * (sh.support_min + (float)n);
        }
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    float *red = s_red + grp * 16;
    if (lane == 0) { red[4 * gw] = m; red[4 * gw + 1] = s0; red[4 * gw + 2] = s1; }
    LZ_HPS(4);
    if (gw != 0) {   // waves 1 and 2 of the head only contribute
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(s_ctr + grp, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }
    while (__hip_atomic_load(s_ctr + grp, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 2) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    LZ_HPS(5);
    {
        const float m1 = red[4], m2 = red[8];
        const float M = fmaxf(fmaxf(m, m1), m2);
        const float e0 = expf(m - M), e1 = expf(m1 - M), e2 = expf(m2 - M);
        const float t0 = (s0 * e0 + red[5] * e1) + red[9] * e2, t1 = (s1 * e0 + red[6] * e1) + red[10] * e2;
        // softmax . support, then InverseScalarTransform.__call__ (scaling_transform.py:82-92) in torch's fp32 op order (lz_hinv.h)
        const float value = t1 / t0;
        const float out = lz_inverse_scalar_transform(value);
        if (sh.dbg_expect && lane == 0) sh.dbg_expect[(size_t)grp * sh.dbg_B + b] = value;   // parity tests (tracing): the pre-transform expectation
        if (lane == 0) {
            (head == 0 ? sh.out_value : sh.out_vp)[off_b + b] = out;
            s_leaf[head == 0 ? 1 : 0] = out;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(s_ctr + 2, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    LZ_HPS(6);
#undef LZ_HPS
}

}  // namespace
