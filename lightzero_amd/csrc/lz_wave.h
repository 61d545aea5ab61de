// lz_wave.h -- wave-wide reductions for the one-wavefront-per-root tree kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace {

// Wave-wide max / min on the DPP path (quad permutes, row mirrors, row broadcasts: ~8 VALU ops) instead of six
// ds_bpermute round trips through the LDS crossbar: these kernels are one wavefront of strictly dependent
// instructions, so every reduction sits on the critical path.  max / min are order independent, the result is exact.
template <bool IS_MAX>
__device__ __forceinline__ float wave_red(float v)
{
    auto op = [](float a, float b) { return IS_MAX ? fmaxf(a, b) : fminf(a, b); };
    int x = __float_as_int(v);
#define LZ_DPP(ctrl, rmask) __int_as_float(__builtin_amdgcn_update_dpp(x, x, ctrl, rmask, 0xf, false))
    v = op(v, LZ_DPP(0xB1, 0xf)); x = __float_as_int(v);   // quad_perm [1,0,3,2]
    v = op(v, LZ_DPP(0x4E, 0xf)); x = __float_as_int(v);   // quad_perm [2,3,0,1]
    v = op(v, LZ_DPP(0x141, 0xf)); x = __float_as_int(v);  // row_half_mirror
    v = op(v, LZ_DPP(0x140, 0xf)); x = __float_as_int(v);  // row_mirror: every lane of a row holds the row result
    v = op(v, LZ_DPP(0x142, 0xa)); x = __float_as_int(v);  // row_bcast:15 into rows 1 and 3
    v = op(v, LZ_DPP(0x143, 0xc));                         // row_bcast:31 into rows 2 and 3: lane 63 holds the total
#undef LZ_DPP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// Wave-wide sum on the same path (fixed pairing order: quads, half rows, rows, then rows 0+1, 2+3, all).  Rows that a
// row_bcast step does not write add 0.
__device__ __forceinline__ float wave_sum(float v)
{
#define LZ_DPP0(ctrl, rmask) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xf, false))
    v += LZ_DPP0(0xB1, 0xf);
    v += LZ_DPP0(0x4E, 0xf);
    v += LZ_DPP0(0x141, 0xf);
    v += LZ_DPP0(0x140, 0xf);
    v += LZ_DPP0(0x142, 0xa);
    v += LZ_DPP0(0x143, 0xc);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// sum over aligned groups of N = 4 | 8 | 16 consecutive lanes (every lane of the group gets it)
template <int N>
__device__ __forceinline__ float group_sum(float v)
{
    static_assert(N == 4 || N == 8 || N == 16, "DPP rows are 16 lanes");
    v += LZ_DPP0(0xB1, 0xf);
    v += LZ_DPP0(0x4E, 0xf);
    if (N >= 8) v += LZ_DPP0(0x141, 0xf);
    if (N >= 16) v += LZ_DPP0(0x140, 0xf);
#undef LZ_DPP0
    return v;
}
// max over lanes 0..15 only (the other rows are ignored), for nodes with at most 16 children: 4 DPP steps instead of 6
__device__ __forceinline__ float row0_max(float v)
{
    int x = __float_as_int(v);
#define LZ_DPPM(ctrl) __int_as_float(__builtin_amdgcn_update_dpp(x, x, ctrl, 0xf, 0xf, false))
    v = fmaxf(v, LZ_DPPM(0xB1)); x = __float_as_int(v);
    v = fmaxf(v, LZ_DPPM(0x4E)); x = __float_as_int(v);
    v = fmaxf(v, LZ_DPPM(0x141)); x = __float_as_int(v);
    v = fmaxf(v, LZ_DPPM(0x140));
#undef LZ_DPPM
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
// v + v[lane ^ 16] and v + v[lane ^ 32] on gfx950's row / half swaps (v_permlane16_swap, v_permlane32_swap): two VALU
// instructions instead of a ds_bpermute round trip; the operands of each addition are the same as with __shfl_xor, so the
// sums are bit-identical
__device__ __forceinline__ float xor16_sum(float v)
{
    const unsigned x = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);  // {r0 r0 r2 r2}, {r1 r1 r3 r3}
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float v)
{
    const unsigned x = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);  // {lo lo}, {hi hi}
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float wave_max(float v) { return wave_red<true>(v); }
__device__ __forceinline__ float wave_min(float v) { return wave_red<false>(v); }

// CMinMaxStats::update (cminimax.cpp:19-26) for a whole chunk of a backup path at once, with the reference's outcome down to the SIGN OF
// ZERO: the reference walks the path and replaces an extremum only on a strict comparison (value > maximum / value < minimum), so
// of several equal extrema the first one in walk order stays -- visible when the equal values are +0 and -0 (all-zero networks in
// two-player mode).  q: this lane's value (walk order = lane order), valid: the lane carries one.  mn / mx: the running statistics
// (wave-uniform), updated in place.
__device__ __forceinline__ void minmax_update_ordered(float q, bool valid, float &mn, float &mx)
{
    const float cmn = wave_min(valid ? q : __builtin_inff()), cmx = wave_max(valid ? q : -__builtin_inff());
    if (cmn < mn) {
        mn = cmn;
        if (cmn == 0.0f) mn = __shfl(q, __builtin_ctzll(__ballot(valid && q == 0.0f)));   // the first zero of the walk, with its sign
    }
    if (cmx > mx) {
        mx = cmx;
        if (cmx == 0.0f) mx = __shfl(q, __builtin_ctzll(__ballot(valid && q == 0.0f)));
    }
}

}  // namespace
