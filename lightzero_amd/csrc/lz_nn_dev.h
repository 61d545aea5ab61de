// lz_nn_dev.h -- device helpers shared by the network translation units (lz_nn.hip, lz_chain_s3g.hip): vector typedefs, the activation
// functions, the exact three-term bf16 split, global-address-space weight pointers, write-through stores, in-graph stamps.
// Include AFTER lz_tree_dev.h (lz_stamp_store, lz_tree_step).
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4;

namespace {

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

}  // namespace
