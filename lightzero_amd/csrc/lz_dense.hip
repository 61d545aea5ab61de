// lz_dense.hip -- dense layers of the vector-observation (MLP) models on the fp32 matrix cores.
//
// Reference layers replaced: MLP_V2 (lzero/model/common.py:28-98) and ding.torch_utils.MLP stacks inside
// RepresentationNetworkMLP (common.py:790-851), PredictionNetworkMLP (common.py:1218-1295 and
// sampled_efficientzero_model_mlp.py), DynamicsNetwork (muzero_model_mlp.py:340-442), DynamicsNetworkMLP
// (efficientzero_model_mlp.py) and ReparameterizationHead, with eval-mode BatchNorm1d / LayerNorm, ReLU / GELU(tanh),
// the residual connection of the dynamics network and InverseScalarTransform (scaling_transform.py:82-92) in the epilogue.
#include "lz_nn_kernels.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float red_sum(float v)  // over the wave that owns the row
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float red_max(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// grid = (ceil(B/16), njobs), block = 1024: 16 waves; wave w owns the 16-column tiles w, w+16, ... in the GEMM and row w
// in the epilogue.  These layers are tiny (<= 0.6 MB of weights, 16 rows): a launch costs a handful of dependent memory
// round trips, not arithmetic.  So everything that does not depend on the GEMM is requested in the first instructions --
// the wave's first 16-k-block chunk of weight fragments (64 VGPRs), the input rows, the residual row, and the epilogue
// vectors (bias / BN / LN) -- and is in flight together; the four waves of a SIMD cover each other's later waits.
constexpr int KCH = 16;   // k-blocks (of 16) per chunk
constexpr int XPT = 8;    // staged input elements per thread: 16 * Kp <= 1024 * XPT  (Kp <= 512)
constexpr int RPT = 10;   // residual / output columns per lane: N <= 64 * RPT

__global__ __launch_bounds__(1024) void k_dense(lz_dense_args a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const lz_dense_job &j = a.job[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r0 = blockIdx.x * 16, B = a.B;
    const int K1 = j.K1, K = j.K1 + j.K2, Kp = (K + 15) & ~15, PS = Kp + 4;
    const int N = j.N, Np = (N + 15) & ~15, NPS = Np + 4;
    const int KB = Kp >> 4, NT = Np >> 4;
    float *sX = smem;              // [16][PS]
    float *sY = sX + 16 * PS;      // [16][NPS]
    float *sP = sY + 16 * NPS;     // [5][Np]: bias, scale, shift, ln_g, ln_b
    const f32x4 *wf4 = reinterpret_cast<const f32x4 *>(j.wf);
    // ---- (1) first weight chunk of this wave
    f32x4 bf[KCH];
    int t = wv, k0 = 0;
    {
        const f32x4 *p = wf4 + (size_t)min(t, NT - 1) * KB * 64 + lane;
#pragma unroll
        for (int q = 0; q < KCH; ++q) bf[q] = p[(size_t)min(q, KB - 1) * 64];
    }
    // ---- (2) epilogue vectors, residual row (row = wave), input rows
    float pv[5] = {0.f, 1.f, 0.f, 1.f, 0.f};
    if (tid < N) {
        pv[0] = j.bias[tid];
        if (j.scale) { pv[1] = j.scale[tid]; pv[2] = j.shift[tid]; }
        if (j.ln_g) { pv[3] = j.ln_g[tid]; pv[4] = j.ln_b[tid]; }
    }
    const int brow = min(r0 + wv, B - 1);
    float rres[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) rres[i] = 0.0f;
    if (j.res) {
        const size_t rbase = (j.res_gather ? (size_t)j.res_gather[brow] * (size_t)j.res_slot_stride : 0) + (size_t)brow * N;
#pragma unroll
        for (int i = 0; i < RPT; ++i) if (lane + 64 * i < N) rres[i] = j.res[rbase + lane + 64 * i];
    }
    float xv[XPT];
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        const int e = tid + 1024 * i;
        xv[i] = 0.0f;
        if (e < 16 * Kp) {
            const int row = e / Kp, k = e - row * Kp;
            const int b = min(r0 + row, B - 1);
            if (k < K1) {
                const size_t base = j.x_gather ? (size_t)j.x_gather[b] * (size_t)j.x_slot_stride : 0;
                xv[i] = j.x[base + (size_t)b * K1 + k];
            } else if (k < K) {
                const int kk = k - K1;
                if (j.x2_mode == 1) xv[i] = j.x2[(size_t)b * j.K2 + kk];
                else if (j.x2_mode == 2) xv[i] = (j.x2_idx[b] == kk) ? 1.0f : 0.0f;
                else xv[i] = (float)j.x2_idx[b] / j.x2_div;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        const int e = tid + 1024 * i;
        if (e < 16 * Kp) { const int row = e / Kp, k = e - row * Kp; sX[row * PS + k] = xv[i]; }
    }
    if (tid < N) {
#pragma unroll
        for (int q = 0; q < 5; ++q) sP[q * Np + tid] = pv[q];
    }
    __syncthreads();
    // ---- (3) GEMM
    const float *sAf = sX + (lane & 15) * PS + (lane >> 4) * 4;
    const int col = lane & 15, rq = 4 * (lane >> 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    while (t < NT) {
#pragma unroll
        for (int q = 0; q < KCH; ++q) {
            if (k0 + q < KB) {
                const f32x4 af = *reinterpret_cast<const f32x4 *>(sAf + (k0 + q) * 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[q][i], acc, 0, 0, 0);
            }
        }
        k0 += KCH;
        if (k0 >= KB) {
#pragma unroll
            for (int q = 0; q < 4; ++q) sY[(rq + q) * NPS + t * 16 + col] = acc[q];
            acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            k0 = 0;
            t += 16;
        }
        if (t < NT) {
            const f32x4 *p = wf4 + (size_t)t * KB * 64 + lane;
#pragma unroll
            for (int q = 0; q < KCH; ++q) bf[q] = p[(size_t)min(k0 + q, KB - 1) * 64];
        }
    }
    __syncthreads();
    // ---- (4) row epilogue: one wave per row
    const int b = r0 + wv;
    const bool live = b < B;
    float *y = sY + wv * NPS;
    float v[RPT];
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int c = lane + 64 * i;
        v[i] = 0.0f;
        if (c < N) {
            v[i] = (y[c] + sP[c]) * sP[Np + c] + sP[2 * Np + c];
            sum += v[i];
        }
    }
    if (j.ln_g) {
        const float mean = red_sum(sum) / (float)N;
        float sq = 0.0f;
#pragma unroll
        for (int i = 0; i < RPT; ++i) if (lane + 64 * i < N) { const float d = v[i] - mean; sq += d * d; }
        const float rstd = 1.0f / sqrtf(red_sum(sq) / (float)N + j.ln_eps);
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int c = lane + 64 * i;
            if (c < N) v[i] = (v[i] - mean) * rstd * sP[3 * Np + c] + sP[4 * Np + c];
        }
    }
    float mx = -__builtin_inff();
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int c = lane + 64 * i;
        if (c < N) {
            float u = v[i];
            if (j.act == 1) u = fmaxf(u, 0.0f);
            else if (j.act == 2) u = 0.5f * u * (1.0f + tanhf(0.7978845608028654f * (u + 0.044715f * u * u * u)));
            u += rres[i];
            if (j.final == 2) {
                if (c >= j.final_split) u = expf(fminf(fmaxf(u, -20.0f), 2.0f));
                else if (j.final_tanh) u = tanhf(u);
            }
            v[i] = u;
            mx = fmaxf(mx, u);
            if (live) {
                if (j.out) j.out[(size_t)b * N + c] = u;
                if (j.out2) j.out2[(size_t)b * N + c] = u;
            }
        }
    }
    if (j.final != 1) return;
    mx = red_max(mx);
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int c = lane + 64 * i;
        if (c < N) {
            const float ex = expf(v[i] - mx);
            s0 += ex;
            s1 += ex * (j.support_min + (float)c);
        }
    }
    s0 = red_sum(s0);
    s1 = red_sum(s1);
    if (lane == 0 && live) {
        // InverseScalarTransform.__call__ (scaling_transform.py:82-92), torch's fp32 op order
        const float value = s1 / s0;
        const float eps = 0.001f;
        float tt = fabsf(value) + 1.0f;
        tt = tt + eps;
        tt = 0.004f * tt;
        tt = 1.0f + tt;
        tt = sqrtf(tt);
        tt = tt - 1.0f;
        tt = tt / 0.002f;
        const float sgn = (value > 0.0f) ? 1.0f : (value < 0.0f ? -1.0f : 0.0f);
        j.out_scalar[b] = sgn * (tt * tt - 1.0f);
    }
}

}  // namespace

void lz_launch_dense(const lz_dense_args &a, hipStream_t s)
{
    size_t lds = 0;
    for (int i = 0; i < a.njobs; ++i) {
        const int Kp = (a.job[i].K1 + a.job[i].K2 + 15) & ~15, Np = (a.job[i].N + 15) & ~15;
        const size_t need = (size_t)(16 * (Kp + 4) + 16 * (Np + 4) + 5 * Np) * 4;
        if (need > lds) lds = need;
    }
    hipLaunchKernelGGL(k_dense, dim3((a.B + 15) / 16, a.njobs), dim3(1024), lds, s, a);
}
