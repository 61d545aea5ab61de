// Micro-benchmark for VERDICT r4 #2: ONE RESIDENT LAUNCH PER SEARCH with XCD-LOCAL groups of 16 workgroups, against today's two
// launches per simulation.  (the only hand-off form of the recurrent loop that had not been measured.)
//
// What is modelled.  A simulation of the search is two dependent phases:
//   A ("chain":  one workgroup = one root; ~30 us of matrix work on a 1.3 MB weight stream shared by every workgroup; publishes the
//                root's row for the LSTM: 1x1 reward-conv output + hidden state = 4.25 KB)
//   B ("LSTM":   a tile of 16 roots; every workgroup of the group reads ALL 16 rows (68 KB) and a weight slice of its own (557 KB),
//                ~10 us; publishes its 16 x 32 block of relu(bn(h')) + head partial sums: 8 KB; phase A of the next simulation
//                reads its root's share of every peer's block: 16 x 512 B)
// Today each phase is a launch (kernel boundary = visibility + dispatch).  Resident form: 256 workgroups (one per CU, 140 KB of LDS)
// stay for all simulations; the 16 workgroups of a group sit on ONE XCD (grouped at run time by HW_REG_XCC_ID, not by block id), so a
// hand-off never leaves that XCD's L2:
//   producer: plain 16-byte stores -> s_waitcnt vmcnt(0) -> __syncthreads -> lane 0: relaxed agent-scope atomic add on the group's flag
//   consumer: lane 0 polls the flag (relaxed agent-scope load = sc1: served by L2) -> __syncthreads -> payload by sc1 loads (bypass the
//             CU's L1, which another CU's stores never refresh; no buffer_inv, no buffer_wbl2: nothing is written back or invalidated)
// mode 0: that.  mode 2: the cross-XCD-safe form (release fence / acquire fence at agent scope) on the same groups.  mode 1: the same
// phase bodies as 2 x S separate launches on one stream (plain loads / stores; the kernel boundary is the hand-off).
// Every consumed word is checked against what its producer must have written (stale reads are COUNTED, under uneven load: the work
// of a workgroup varies by +-25 % with its id and the simulation).
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/ubench/libxcdresident.so tools/ubench/xcd_resident.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

struct ctl_t {
    unsigned xcc_count[16];      // arrivals per XCC id (grouping)
    unsigned fault;              // spin limit hit / more than 32 workgroups on one XCD
    unsigned stale;              // consumed words that were not what the producer wrote
    unsigned group_of[256], member_of[256];   // filled by the resident kernel; reused by the launch-per-phase form
};

typedef float v4f __attribute__((ext_vector_type(4)));

// 16-byte load that bypasses the CU's vector L1 (sc1: served by the XCD's L2), through a buffer descriptor so that the compiler
// tracks its completion like any other load (several stay in flight)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f load_sc1(const v4f *base, size_t bytes, unsigned byte_off)
{
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, 0x00020000);
    const v4u u = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 16);
    return __builtin_bit_cast(v4f, u);
}
__device__ __forceinline__ float tag(int sim, int phase, int producer, int i) { return (float)((sim * 2 + phase) * 4096 + producer * 16 + (i & 15)) + 0.5f; }

// the "matrix work" of a phase: stream `stream_v4` 16-byte vectors of weights through the CU (L2-resident after the first pass) and spin
// on LDS until `cycles` have passed
__device__ __forceinline__ float work(const v4f *w, int stream_v4, int cycles, float *smem)
{
    const int tid = threadIdx.x;
    const unsigned long long c0 = __builtin_readcyclecounter();
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < stream_v4; i += 512 * 4) {
        v4f a = w[i], b = w[min(i + 512, stream_v4 - 1)], c = w[min(i + 1024, stream_v4 - 1)], d = w[min(i + 1536, stream_v4 - 1)];
        acc += a + b + c + d;
    }
    float x = acc.x + acc.y + acc.z + acc.w;
    while ((long long)(__builtin_readcyclecounter() - c0) < cycles) smem[tid] = x * 1.0001f + smem[(tid + 1) & 511];
    return x + smem[tid];
}

struct args_t {
    ctl_t *ctl;
    unsigned *flagA, *flagB;          // [32 groups][sims]
    v4f *payA, *payB;                 // [256][VA] / [256][VB] 16-byte vectors per workgroup (double-buffered by simulation parity)
    const v4f *wA, *wB;               // weight streams
    unsigned long long *stamps;       // [sims][256][4]: A ready, A end, B ready, B end (100 MHz)
    int sims, cyclesA, cyclesB, streamA, streamB, mode, phase_only, sim0;
};
constexpr int VA = 272, VB = 512;     // 4.25 KB and 8 KB per workgroup, in 16-byte vectors

__device__ __forceinline__ bool wait_flag(unsigned *f, unsigned want, ctl_t *ctl)
{
    int spins = 0;
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { atomicAdd(&ctl->fault, 1u); return false; }
    }
    return true;
}

// phase bodies.  resident: mode 0 (XCD-local) / 2 (agent fences); launches: mode 1
__device__ __forceinline__ void phase_A(const args_t &a, int s, int grp, int mem, int wg, float *smem, unsigned long long *st)
{
    const int tid = threadIdx.x, par = s & 1;
    float chk = 0.0f;
    if (s > 0) {   // this root's share of every peer's phase-B block of the previous simulation: 16 x 512 B
        if (a.mode != 1) {
            if (tid == 0) wait_flag(a.flagB + (size_t)grp * a.sims + (s - 1), 16u, a.ctl);
            __syncthreads();
            if (a.mode == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        if (tid == 0) st[0] = __builtin_amdgcn_s_memrealtime();
        const int peer = tid >> 5, i = tid & 31;   // 16 peers x 32 vectors
        const size_t idx = ((size_t)((par ^ 1) * 256 + grp * 16 + peer)) * VB + mem * 32 + i;
        const v4f v = (a.mode == 0) ? load_sc1(a.payB, (size_t)2 * 256 * VB * 16, (unsigned)(idx * 16)) : a.payB[idx];
        const float want = tag(s - 1, 1, peer, mem * 32 + i);
        if (v.x != want || v.w != want) atomicAdd(&a.ctl->stale, 1u);
        chk = v.y;
    } else if (tid == 0) st[0] = __builtin_amdgcn_s_memrealtime();
    const int jitter = a.cyclesA / 4 * (((wg * 7 + s * 3) % 9) - 4) / 4;    // uneven load: +-25 %
    chk += work(a.wA, a.streamA, a.cyclesA + jitter, smem);
    if (tid < VA) {
        const float t = tag(s, 0, mem, tid);
        v4f v = {t, chk * 0.0f, 0.0f, t};
        a.payA[((size_t)(par * 256 + grp * 16 + mem)) * VA + tid] = v;
    }
    if (a.mode != 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (a.mode == 2) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            __hip_atomic_fetch_add(a.flagA + (size_t)grp * a.sims + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid == 0) st[1] = __builtin_amdgcn_s_memrealtime();
}

__device__ __forceinline__ void phase_B(const args_t &a, int s, int grp, int mem, int wg, float *smem, unsigned long long *st)
{
    const int tid = threadIdx.x, par = s & 1;
    if (a.mode != 1) {
        if (tid == 0) wait_flag(a.flagA + (size_t)grp * a.sims + s, 16u, a.ctl);
        __syncthreads();
        if (a.mode == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    if (tid == 0) st[2] = __builtin_amdgcn_s_memrealtime();
    float chk = 0.0f;
    for (int k = tid; k < 16 * VA; k += 512) {   // all 16 rows of the group: 68 KB
        const int peer = k / VA, i = k - peer * VA;
        const size_t idx = ((size_t)(par * 256 + grp * 16 + peer)) * VA + i;
        const v4f v = (a.mode == 0) ? load_sc1(a.payA, (size_t)2 * 256 * VA * 16, (unsigned)(idx * 16)) : a.payA[idx];
        const float want = tag(s, 0, peer, i);
        if (v.x != want || v.w != want) atomicAdd(&a.ctl->stale, 1u);
        chk += v.y;
    }
    const int jitter = a.cyclesB / 4 * (((wg * 5 + s * 7) % 9) - 4) / 4;
    chk += work(a.wB + (size_t)wg * a.streamB, a.streamB, a.cyclesB + jitter, smem);
    {
        const float t = tag(s, 1, mem, tid);
        v4f v = {t, chk * 0.0f, 0.0f, t};
        a.payB[((size_t)(par * 256 + grp * 16 + mem)) * VB + tid] = v;   // 512 vectors = 8 KB
    }
    if (a.mode != 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (a.mode == 2) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            __hip_atomic_fetch_add(a.flagB + (size_t)grp * a.sims + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid == 0) st[3] = __builtin_amdgcn_s_memrealtime();
}

extern "C" __global__ __launch_bounds__(512) void k_resident(args_t a)
{
    extern __shared__ float smem[];
    __shared__ int s_grp, s_mem;
    const int wg = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        // group by the XCD the workgroup really runs on: 16 arrivals of one XCC id form a group
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xf;
        const unsigned slot = atomicAdd(&a.ctl->xcc_count[xcc], 1u);
        if (slot >= 32 || xcc >= 8) atomicAdd(&a.ctl->fault, 1u);
        s_grp = (int)((xcc & 7) * 2 + (slot >> 4) % 2);
        s_mem = (int)(slot & 15);
        a.ctl->group_of[wg] = (unsigned)s_grp;
        a.ctl->member_of[wg] = (unsigned)s_mem;
    }
    __syncthreads();
    const int grp = s_grp, mem = s_mem;
    smem[tid] = 0.0f;
    for (int s = 0; s < a.sims; ++s) {
        if (__hip_atomic_load(&a.ctl->fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;   // somebody gave up waiting: leave, do not spin on
        unsigned long long st[4] = {0, 0, 0, 0};
        phase_A(a, s, grp, mem, wg, smem, st);
        phase_B(a, s, grp, mem, wg, smem, st);
        if (tid == 0)
            for (int i = 0; i < 4; ++i) a.stamps[((size_t)s * 256 + wg) * 4 + i] = st[i];
    }
}

extern "C" __global__ __launch_bounds__(512) void k_phase(args_t a)   // one phase of one simulation as its own launch (mode 1)
{
    extern __shared__ float smem[];
    const int wg = blockIdx.x, tid = threadIdx.x;
    const int grp = (int)a.ctl->group_of[wg], mem = (int)a.ctl->member_of[wg];
    smem[tid] = 0.0f;
    unsigned long long st[4] = {0, 0, 0, 0};
    if (a.phase_only == 0) phase_A(a, a.sim0, grp, mem, wg, smem, st);
    else phase_B(a, a.sim0, grp, mem, wg, smem, st);
    if (tid == 0) {
        const int b = a.phase_only * 2;
        a.stamps[((size_t)a.sim0 * 256 + wg) * 4 + b] = st[b];
        a.stamps[((size_t)a.sim0 * 256 + wg) * 4 + b + 1] = st[b + 1];
    }
}

// returns hipError; *out_ms = host wall time of the whole run (launch to completion)
extern "C" int xcd_resident_run(ctl_t *ctl, unsigned *flagA, unsigned *flagB, void *payA, void *payB, const void *wA, const void *wB,
                                unsigned long long *stamps, int sims, int cyclesA, int cyclesB, int streamA, int streamB, int mode,
                                int lds_bytes, int cooperative, float *out_ms)
{
    args_t a{ctl, flagA, flagB, (v4f *)payA, (v4f *)payB, (const v4f *)wA, (const v4f *)wB, stamps, sims, cyclesA, cyclesB, streamA, streamB, mode, 0, 0};
    hipMemset(flagA, 0, (size_t)32 * sims * 4);
    hipMemset(flagB, 0, (size_t)32 * sims * 4);
    if (mode != 1) hipMemset(ctl, 0, sizeof(unsigned) * 18);   // grouping + fault / stale counters (mode 1 reuses the groups of the last resident run)
    else hipMemset(&ctl->fault, 0, 8);
    hipFuncSetAttribute((const void *)k_resident, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipFuncSetAttribute((const void *)k_phase, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipError_t err = hipSuccess;
    if (mode != 1) {
        if (cooperative) {
            void *params[] = {&a};
            err = hipLaunchCooperativeKernel((const void *)k_resident, dim3(256), dim3(512), params, (unsigned)lds_bytes, 0);
        } else {
            hipLaunchKernelGGL(k_resident, dim3(256), dim3(512), lds_bytes, 0, a);
        }
    } else {
        for (int s = 0; s < sims; ++s) {
            a.sim0 = s;
            a.phase_only = 0; hipLaunchKernelGGL(k_phase, dim3(256), dim3(512), lds_bytes, 0, a);
            a.phase_only = 1; hipLaunchKernelGGL(k_phase, dim3(256), dim3(512), lds_bytes, 0, a);
        }
    }
    hipEventRecord(e1, 0);
    hipError_t e2 = hipDeviceSynchronize();
    if (err == hipSuccess) err = e2;
    if (out_ms) hipEventElapsedTime(out_ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return (int)err;
}
