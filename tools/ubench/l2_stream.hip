// Micro-benchmark: how fast can ONE workgroup per CU stream a weight buffer that EVERY CU reads (the access pattern of the
// convolution chain, the LSTM and the head kernels: 256 workgroups, all reading the same few hundred KB from L2)?
//   mode 0: every workgroup reads the same addresses in the same order      mode 1: workgroup b starts at a rotated offset
//   mode 2: every workgroup reads its own private copy (no sharing)
// Each wave keeps DEPTH wave-wide 1 KB loads (global_load_dwordx4, 16 B / lane) in flight.  Build: hipcc -O3 --offload-arch=gfx950
// -shared -fPIC -o libl2stream.so l2_stream.hip.  Driver: tools/ubench/l2_stream.py
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(512) void k_stream(const f32x4 *__restrict__ w, int kb_per_wave, int total_kb, int mode, int reps, float *out,
                                                unsigned long long *cycles)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, b = blockIdx.x;
    const int nw = blockDim.x >> 6;
    // wave wv of the workgroup owns the KB range [wv, wv + nw, ...) like the chain's N-tiles
    size_t base = 0;
    if (mode == 2) base = (size_t)b * total_kb;   // private copy, in KB
    const int rot = mode == 1 ? (b * 37) % kb_per_wave : 0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        f32x4 q[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int i = (d + rot) % kb_per_wave;
            q[d] = w[(base + (size_t)(i * nw + wv)) * 64 + lane];
        }
        for (int i0 = 0; i0 < kb_per_wave; i0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                acc += q[d];
                const int nxt = i0 + DEPTH + d;
                const int i = (min(nxt, kb_per_wave - 1) + rot) % kb_per_wave;
                q[d] = w[(base + (size_t)(i * nw + wv)) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += q[d];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && wv == 0) cycles[b] = t1 - t0;
    if (acc[0] == 123.456f) out[0] = acc[1];
}

extern "C" int l2stream_run(const void *d_w, int kb_per_wave, int total_kb, int mode, int reps, int depth, int blocks, int threads, void *d_out,
                            void *d_cycles, float *ms)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
#define RUN(D) hipLaunchKernelGGL((k_stream<D>), dim3(blocks), dim3(threads), 0, 0, (const f32x4 *)d_w, kb_per_wave, total_kb, mode, reps, (float *)d_out, (unsigned long long *)d_cycles)
    if (depth == 4) RUN(4); else if (depth == 8) RUN(8); else if (depth == 12) RUN(12); else if (depth == 16) RUN(16); else if (depth == 24) RUN(24); else RUN(32);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(ms, e0, e1);
    return (int)hipGetLastError();
}
