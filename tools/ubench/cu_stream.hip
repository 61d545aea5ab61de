// Micro-benchmark: what limits one CU's weight stream?  256 workgroups x NW waves, every wave streams its own region of a buffer that
// all workgroups share (the access pattern of k_chain_w: same addresses on every CU, separate 64 KB streams per wave).
//   mode 0: global_load_dwordx4 from the L2-resident buffer            mode 1: the same with the non-temporal hint
//   mode 2: every wave re-reads one 4 KB window (L1-resident: the TA -> VGPR return path alone)
//   mode 3: global_load_lds_dwordx4 (DMA into LDS, no VGPR return), L2-resident buffer
//   mode 4: ds_read_b128 only (LDS -> VGPR), for scale        modes 5 / 6 / 7: ds_write_b128 / b64 / b32 only (VGPR -> LDS)
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o libcustream.so cu_stream.hip.  Driver: tools/ubench/cu_stream.py
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void k_cu_stream(const f32x4 *__restrict__ w, int kb_per_wave, int reps, float *out, unsigned long long *cycles)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const f32x4 *base = w + (size_t)wv * kb_per_wave * 64 + lane;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 *lds = reinterpret_cast<f32x4 *>(smem) + threadIdx.x;   // one float4 slot per thread, DEPTH slots apart by blockDim
    if (MODE == 4) for (int d = 0; d < DEPTH; ++d) lds[d * blockDim.x] = (f32x4){1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        if (MODE == 3) {
#if __has_builtin(__builtin_amdgcn_global_load_lds)
            for (int i = 0; i < kb_per_wave; ++i) {
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(base + (size_t)i * 64 - lane),
                                                 (void __attribute__((address_space(3))) *)(reinterpret_cast<f32x4 *>(smem) + (size_t)((i % DEPTH) * blockDim.x + wv * 64)), 16, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0);
#endif
        } else if (MODE == 4) {
            for (int i = 0; i < kb_per_wave; ++i) acc += lds[(i % DEPTH) * blockDim.x];
        } else if (MODE == 5) {   // 1 KB per wave and store
            for (int i = 0; i < kb_per_wave; ++i) { lds[(i % DEPTH) * blockDim.x] = acc; acc[0] += 1.0f; }
        } else if (MODE == 6) {   // 512 B per wave and store: twice the stores for the same bytes
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 *l2 = reinterpret_cast<f32x2 *>(smem) + threadIdx.x;
            for (int i = 0; i < 2 * kb_per_wave; ++i) { l2[(i % DEPTH) * blockDim.x] = (f32x2){acc[0], acc[1]}; acc[0] += 1.0f; }
        } else if (MODE == 7) {
            float *l1 = smem + threadIdx.x;
            for (int i = 0; i < 4 * kb_per_wave; ++i) { l1[(i % DEPTH) * blockDim.x] = acc[0]; acc[0] += 1.0f; }
        } else {
            f32x4 q[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const f32x4 *p = base + (size_t)(MODE == 2 ? (d & 3) : d) * 64;
                q[d] = MODE == 1 ? __builtin_nontemporal_load(p) : *p;
            }
            for (int i0 = 0; i0 < kb_per_wave; i0 += DEPTH) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    acc += q[d];
                    const int nxt = min(i0 + DEPTH + d, kb_per_wave - 1);
                    const f32x4 *p = base + (size_t)(MODE == 2 ? (nxt & 3) : nxt) * 64;
                    q[d] = MODE == 1 ? __builtin_nontemporal_load(p) : *p;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc += q[d];
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && wv == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc[0] == 123.456f) out[0] = acc[1] + smem[threadIdx.x];
}

extern "C" int custream_run(const void *d_w, int kb_per_wave, int mode, int reps, int blocks, int threads, void *d_out, void *d_cycles, float *ms)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = (size_t)16 * threads * 16;
    hipEventRecord(e0, 0);
#define RUN(M) hipLaunchKernelGGL((k_cu_stream<M, 16>), dim3(blocks), dim3(threads), lds, 0, (const f32x4 *)d_w, kb_per_wave, reps, (float *)d_out, (unsigned long long *)d_cycles)
    if (mode == 0) RUN(0); else if (mode == 1) RUN(1); else if (mode == 2) RUN(2); else if (mode == 3) RUN(3); else if (mode == 4) RUN(4);
    else if (mode == 5) RUN(5); else if (mode == 6) RUN(6); else RUN(7);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(ms, e0, e1);
    return (int)hipGetLastError();
}
