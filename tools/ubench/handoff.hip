// Micro-benchmark: what does a PHASE HANDOFF cost inside one launch, against a kernel boundary?
// A launch of P phases x 256 workgroups (140 KB of LDS each: one per CU).  A workgroup of phase p "works" for `work_cycles`, writes a
// line of data, releases (agent scope) and bumps the phase counter of its group of 32 (ids = group mod 8: the workgroups of one XCD); a
// workgroup of phase p + 1 waits -- relaxed polls, then one acquire -- until its group's counter of phase p has reached 32, then reads what
// the 32 producers wrote.  It only ever waits for LOWER block ids (in-order dispatch: no co-residency assumption).  Every workgroup stamps
// s_memrealtime at its start, when its wait ends and at its end.  The driver compares the phase period with the same work as P separate
// launches on one stream.   hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o libhandoff.so handoff.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" __global__ __launch_bounds__(512) void k_phases(unsigned *done, float *data, unsigned long long *stamps, int work_cycles, int phase0, int use_wait,
                                                           unsigned *fault)
{
    extern __shared__ float smem[];
    const int wg = blockIdx.x, tid = threadIdx.x;
    const int phase = phase0 + wg / 256, idx = wg % 256, grp = idx % 8;
    unsigned long long t0 = 0, t1 = 0;
    if (tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
    if (use_wait && phase > 0) {
        if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(done + (phase - 1) * 8 + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 32u) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 20)) { *fault = 1; break; }
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    if (tid == 0) t1 = __builtin_amdgcn_s_memrealtime();
    // consume what the 32 producers of the previous phase wrote (one 2 KB line each), then "work"
    float acc = 0.0f;
    if (phase > 0) {
        for (int k = 0; k < 32; ++k) acc += data[((size_t)(phase - 1) * 256 + k * 8 + grp) * 512 + tid];
    }
    const unsigned long long c0 = __builtin_readcyclecounter();
    while ((long long)(__builtin_readcyclecounter() - c0) < work_cycles) smem[tid] = acc * 1.0001f + smem[(tid + 1) & 511];
    data[((size_t)phase * 256 + idx) * 512 + tid] = acc + smem[tid] + 1.0f;
    __syncthreads();
    if (tid == 0) {
        if (use_wait) __hip_atomic_fetch_add(done + phase * 8 + grp, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        stamps[(size_t)(phase * 256 + idx) * 3 + 0] = t0;
        stamps[(size_t)(phase * 256 + idx) * 3 + 1] = t1;
        stamps[(size_t)(phase * 256 + idx) * 3 + 2] = __builtin_amdgcn_s_memrealtime();
    }
}

extern "C" int handoff_run(unsigned *done, float *data, unsigned long long *stamps, unsigned *fault, int phases, int work_cycles, int one_launch, int lds_bytes)
{
    hipMemset(done, 0, (size_t)phases * 8 * 4);
    hipDeviceSynchronize();
    if (one_launch) {
        hipLaunchKernelGGL(k_phases, dim3(phases * 256), dim3(512), lds_bytes, 0, done, data, stamps, work_cycles, 0, 1, fault);
    } else {
        for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(k_phases, dim3(256), dim3(512), lds_bytes, 0, done, data, stamps, work_cycles, p, 0, fault);
    }
    return (int)hipDeviceSynchronize();
}
