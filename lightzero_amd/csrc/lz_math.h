// lz_math.h -- binary32 expf / logf with results identical to glibc's libm, usable from HIP device
// code and from host C/C++.
//
// Why: the reference tree computes its priors with exp(float) and its pUCT bonus with log(float)
// (lzero/mcts/ctree/ctree_efficientzero/lib/cnode.cpp:134, :776), i.e. the HOST libm's expf/logf.
// Visit counts are decided by float comparisons of those values, so the device tree must produce
// the same bits.  glibc (>= 2.27; 2.35 in this image) implements both functions with the
// table + double-precision polynomial algorithm of ARM's optimized-routines (Szabolcs Nagy, 2017;
// sysdeps/ieee754/flt-32/e_expf.c, e_logf.c, e_exp2f_data.c, e_logf_data.c).  This header restates
// that published algorithm with the same constants.  Operations are individually rounded IEEE
// binary64 operations (compile with -ffp-contract=off) except where an explicit fma() reproduces
// glibc's x86-64 FMA ifunc build.  tests/test_lz_math.py checks bit-equality with the host libm
// over EVERY binary32 input (2^32 values, both functions).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define LZ_HD __host__ __device__ __forceinline__
#else
#define LZ_HD static inline
#endif

LZ_HD uint32_t lz_asuint(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
LZ_HD float lz_asfloat(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }
LZ_HD uint64_t lz_asuint64(double d) { union { double d; uint64_t u; } c; c.d = d; return c.u; }
LZ_HD double lz_asdouble(uint64_t u) { union { double d; uint64_t u; } c; c.u = u; return c.d; }

// 2^(i/32) tables: T[i] = asuint64(2^(i/32)) - (i << 47)
LZ_HD uint64_t lz_exp2f_tab(unsigned i)
{
    const uint64_t T[32] = {
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
        0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
        0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
        0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
        0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
        0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
        0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
        0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
    };
    return T[i];
}

// tab: optional copy of the 32-entry table somewhere faster than the constant segment (LDS); NULL = lz_exp2f_tab
LZ_HD float lz_expf_core(float x, const uint64_t *tab)
{
    const double InvLn2N = 0x1.71547652b82fep+5;  // 32/ln2
    const double SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-20, C1 = 0x1.ebfce50fac4f3p-13, C2 = 0x1.62e42ff0c52d6p-6;
    const uint32_t ux = lz_asuint(x);
    const uint32_t abstop = (ux >> 20) & 0x7ff;
    if (abstop >= 0x42b) {  // |x| >= 88 or nan/inf   (top12(88.0f) = 0x42b)
        if (ux == 0xff800000u) return 0.0f;            // -inf
        if (abstop >= 0x7f8) return x + x;             // +inf / nan
        if (x > 0x1.62e42ep6f) return lz_asfloat(0x7f800000u);   // overflow
        if (x < -0x1.9fe368p6f) return 0.0f;                     // underflow
    }
    const double xd = (double)x;
    double z = InvLn2N * xd;
    double kd = z + SHIFT;
    const uint64_t ki = lz_asuint64(kd);
    kd -= SHIFT;
    // glibc selects its FMA build of expf on every x86-64 CPU that has FMA (ifunc); that build
    // contracts r and the polynomial.  The contracted form below is the one that matches it on all
    // 2^32 inputs; the uncontracted form differs for 2 inputs near x = -63.1 (tests/test_lz_math.py).
    const double r = __builtin_fma(InvLn2N, xd, -kd);
    uint64_t t = tab ? tab[ki & 31] : lz_exp2f_tab((unsigned)(ki & 31));
    t += ki << 47;
    const double s = lz_asdouble(t);
    z = __builtin_fma(C0, r, C1);
    const double r2 = r * r;
    double y = __builtin_fma(C2, r, 1.0);
    y = __builtin_fma(z, r2, y);
    y = y * s;
    return (float)y;
}
LZ_HD float lz_expf(float x) { return lz_expf_core(x, 0); }

LZ_HD float lz_logf(float x)
{
    const double invc[16] = {
        0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0,  0x1.3c995b0b80385p+0,
        0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,  0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0,
        0x1.0953f419900a7p+0, 0x1p+0,               0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
        0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1,
    };
    const double logc[16] = {
        -0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3,
        -0x1.6574f0ac07758p-3, -0x1.1aa2bc79c81p-3,   -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4,
        -0x1.252f438e10c1ep-5, 0x0p+0,                0x1.aa5aa5df25984p-5,  0x1.c5e53aa362eb4p-4,
        0x1.526e57720db08p-3,  0x1.bc2860d22477p-3,   0x1.1058bc8a07ee1p-2,  0x1.4043057b6ee09p-2,
    };
    const double Ln2 = 0x1.62e42fefa39efp-1;
    const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    uint32_t ix = lz_asuint(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2 == 0) return lz_asfloat(0xff800000u);           // log(+-0) = -inf
        if (ix == 0x7f800000u) return x;                           // log(inf) = inf
        if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return lz_asfloat(0x7fc00000u);  // x < 0 or nan
        ix = lz_asuint(x * 0x1p23f);                               // subnormal: normalise
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15);
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    const double z = (double)lz_asfloat(iz);
    const double r = z * invc[i] - 1.0;
    const double y0 = logc[i] + (double)k * Ln2;
    const double r2 = r * r;
    double y = A1 * r + A2;
    y = A0 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}
