// Micro-benchmark: the products loop of k_chain_w in isolation -- per step one 1 KB weight fragment per wave, one 1 KB LDS read (the A
// fragment) and 12 v_mfma_f32_4x4x1 -- with the fragment arriving (mode 0) in registers (global_load_dwordx4, ring of 16) or (mode 1) in
// LDS by DMA (global_load_lds_dwordx4, ring of 8 slots per wave) and read back with ds_read_b128.  Question: is the ~34 B/clk per CU
// that the chain and the LSTM sustain under MFMA load a limit of the path that returns vector-memory data to the register files, and
// does the DMA path avoid it?   mode 2: no weight loads at all (the matrix + LDS floor).
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o libmfmastream.so mfma_stream.hip.  Driver: tools/ubench/mfma_stream.py
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k_mfma_stream(const f32x4 *__restrict__ w, int steps, int reps, float *out, unsigned long long *cycles)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
    float *sA = smem;                                  // 16 KB of "transformed patches"
    float *sW = smem + 4096 + wv * 8 * 256;            // mode 1: this wave's 8 slots of 1 KB
    for (int i = tid; i < 4096; i += blockDim.x) sA[i] = (float)(i & 7);
    __syncthreads();
    const f32x4 *base = w + (size_t)wv * steps * 64 + lane;   // a contiguous region per wave, the same on every CU
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0) {
            f32x4 q[16];
#pragma unroll
            for (int d = 0; d < 16; ++d) q[d] = base[(size_t)d * 64];
            for (int s0 = 0; s0 < steps; s0 += 16) {
#pragma unroll
                for (int d = 0; d < 16; ++d) {
                    const f32x4 b = q[d];
                    q[d] = base[(size_t)min(s0 + 16 + d, steps - 1) * 64];
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(sA + ((s0 + d) & 255) * 16 + (lane & 15) * 4);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j], b[j], acc0, 4, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j], b[j], acc1, 4, 1, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j], b[j], acc2, 4, 2, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int d = 0; d < 8; ++d)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (size_t)d * 64),
                                                 (__attribute__((address_space(3))) void *)(sW + d * 256), 16, 0, 0);
            for (int s0 = 0; s0 < steps; s0 += 8) {
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");   // the oldest of the 8 DMA loads has landed
                    const f32x4 b = *reinterpret_cast<const f32x4 *>(sW + d * 256 + lane * 4);
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(sA + ((s0 + d) & 255) * 16 + (lane & 15) * 4);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot is read before it is overwritten
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (size_t)min(s0 + 8 + d, steps - 1) * 64),
                                                     (__attribute__((address_space(3))) void *)(sW + d * 256), 16, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j], b[j], acc0, 4, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j], b[j], acc1, 4, 1, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j], b[j], acc2, 4, 2, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            const f32x4 b = base[0];
            for (int s0 = 0; s0 < steps; s0 += 16) {
#pragma unroll
                for (int d = 0; d < 16; ++d) {
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(sA + ((s0 + d) & 255) * 16 + (lane & 15) * 4);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j], b[j], acc0, 4, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j], b[j], acc1, 4, 1, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j], b[j], acc2, 4, 2, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc0[0] + acc1[1] + acc2[2] == 123.456f) out[0] = acc0[1];
    (void)nw;
}

extern "C" int mfmastream_run(const void *d_w, int steps, int mode, int reps, int blocks, int threads, void *d_out, void *d_cycles, float *ms)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t lds = (size_t)(4096 + (threads / 64) * 8 * 256) * 4;
    (void)hipEventRecord(e0, 0);
#define RUN(M) hipLaunchKernelGGL((k_mfma_stream<M>), dim3(blocks), dim3(threads), lds, 0, (const f32x4 *)d_w, steps, reps, (float *)d_out, (unsigned long long *)d_cycles)
    if (mode == 0) RUN(0); else if (mode == 1) RUN(1); else RUN(2);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(ms, e0, e1);
    return (int)hipGetLastError();
}
