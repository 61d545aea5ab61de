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

__device__ __forceinline__ float red16_sum(float v)
{
    v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    return v;
}
__device__ __forceinline__ float red16_max(float v)
{
    v = fmaxf(v, __shfl_xor(v, 8)); v = fmaxf(v, __shfl_xor(v, 4)); v = fmaxf(v, __shfl_xor(v, 2)); v = fmaxf(v, __shfl_xor(v, 1));
    return v;
}

// grid = (ceil(B/16), njobs), block = 256 (4 waves: wave w owns the 16-column tiles w, w+4, ...).
__global__ __launch_bounds__(256) void k_dense(lz_dense_args a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const lz_dense_job &j = a.job[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r0 = blockIdx.x * 16, B = a.B;
    const int K1 = j.K1, K = j.K1 + j.K2, Kp = (K + 15) & ~15, PS = Kp + 4;
    const int N = j.N, Np = (N + 15) & ~15, NPS = Np + 4;
    const int KB = Kp >> 4, NT = Np >> 4;
    float *sX = smem;             // [16][PS]
    float *sY = smem + 16 * PS;   // [16][NPS]
    // ---- stage the 16 input rows (gathered from a pool slot, action encoding appended, zero padded)
    for (int i = tid; i < 16 * Kp; i += 256) {
        const int row = i / Kp, k = i - row * Kp;
        const int b = min(r0 + row, B - 1);
        float v = 0.0f;
        if (k < K1) {
            const size_t base = j.x_gather ? (size_t)j.x_gather[b] * (size_t)j.x_slot_stride : 0;
            v = j.x[base + (size_t)b * K1 + k];
        } else if (k < K) {
            const int kk = k - K1;
            if (j.x2_mode == 1) v = j.x2[(size_t)b * j.K2 + kk];
            else if (j.x2_mode == 2) v = (j.x2_idx[b] == kk) ? 1.0f : 0.0f;
            else v = (float)j.x2_idx[b] / j.x2_div;
        }
        sX[row * PS + k] = v;
    }
    __syncthreads();
    // ---- GEMM: 4 column tiles per wave and pass; the weight fragments of k-block kb + 1 are requested before the
    // 16 MFMAs of k-block kb issue
    const f32x4 *wf4 = reinterpret_cast<const f32x4 *>(j.wf);
    const float *sAf = sX + (lane & 15) * PS + (lane >> 4) * 4;
    for (int t0 = wv; t0 < NT; t0 += 16) {
        const int tA = t0, tB = min(t0 + 4, NT - 1), tC = min(t0 + 8, NT - 1), tD = min(t0 + 12, NT - 1);
        const f32x4 *pA = wf4 + (size_t)tA * KB * 64 + lane, *pB = wf4 + (size_t)tB * KB * 64 + lane;
        const f32x4 *pC = wf4 + (size_t)tC * KB * 64 + lane, *pD = wf4 + (size_t)tD * KB * 64 + lane;
        f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = accA, accC = accA, accD = accA;
        f32x4 bA = pA[0], bB = pB[0], bC = pC[0], bD = pD[0];
        for (int kb = 0; kb < KB; ++kb) {
            const int kn = min(kb + 1, KB - 1);
            const f32x4 nA = pA[kn * 64], nB = pB[kn * 64], nC = pC[kn * 64], nD = pD[kn * 64];
            const f32x4 af = *reinterpret_cast<const f32x4 *>(sAf + kb * 16);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                accA = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q], bA[q], accA, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q], bB[q], accB, 0, 0, 0);
                accC = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q], bC[q], accC, 0, 0, 0);
                accD = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q], bD[q], accD, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            bA = nA; bB = nB; bC = nC; bD = nD;
        }
        const int col = lane & 15, rq = 4 * (lane >> 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sY[(rq + q) * NPS + tA * 16 + col] = accA[q];
            if (t0 + 4 < NT) sY[(rq + q) * NPS + tB * 16 + col] = accB[q];
            if (t0 + 8 < NT) sY[(rq + q) * NPS + tC * 16 + col] = accC[q];
            if (t0 + 12 < NT) sY[(rq + q) * NPS + tD * 16 + col] = accD[q];
        }
    }
    __syncthreads();
    // ---- row epilogue: 16 lanes per row
    const int row = tid >> 4, li = tid & 15;
    const int b = r0 + row;
    const bool live = b < B;
    float *y = sY + row * NPS;
    float sum = 0.0f;
    for (int c = li; c < N; c += 16) {
        float v = y[c] + j.bias[c];
        if (j.scale) v = v * j.scale[c] + j.shift[c];
        y[c] = v;
        sum += v;
    }
    float mean = 0.0f, rstd = 1.0f;
    if (j.ln_g) {
        mean = red16_sum(sum) / (float)N;
        float sq = 0.0f;
        for (int c = li; c < N; c += 16) { const float d = y[c] - mean; sq += d * d; }
        rstd = 1.0f / sqrtf(red16_sum(sq) / (float)N + j.ln_eps);
    }
    const size_t rbase = j.res ? ((j.res_gather ? (size_t)j.res_gather[min(b, B - 1)] * (size_t)j.res_slot_stride : 0) + (size_t)min(b, B - 1) * N) : 0;
    float mx = -__builtin_inff();
    for (int c = li; c < N; c += 16) {
        float v = y[c];
        if (j.ln_g) v = (v - mean) * rstd * j.ln_g[c] + j.ln_b[c];
        if (j.act == 1) v = fmaxf(v, 0.0f);
        else if (j.act == 2) v = 0.5f * v * (1.0f + tanhf(0.7978845608028654f * (v + 0.044715f * v * v * v)));
        if (j.res) v += j.res[rbase + c];
        if (j.final == 2) {
            if (c >= j.final_split) v = expf(fminf(fmaxf(v, -20.0f), 2.0f));
            else if (j.final_tanh) v = tanhf(v);
        }
        y[c] = v;
        mx = fmaxf(mx, v);
        if (live) {
            if (j.out) j.out[(size_t)b * N + c] = v;
            if (j.out2) j.out2[(size_t)b * N + c] = v;
        }
    }
    if (j.final != 1) return;
    mx = red16_max(mx);
    float s0 = 0.0f, s1 = 0.0f;
    for (int c = li; c < N; c += 16) {
        const float ex = expf(y[c] - mx);
        s0 += ex;
        s1 += ex * (j.support_min + (float)c);
    }
    s0 = red16_sum(s0);
    s1 = red16_sum(s1);
    if (li == 0 && live) {
        // InverseScalarTransform.__call__ (scaling_transform.py:82-92), torch's fp32 op order
        const float value = s1 / s0;
        const float eps = 0.001f;
        float t = fabsf(value) + 1.0f;
        t = t + eps;
        t = 0.004f * t;
        t = 1.0f + t;
        t = sqrtf(t);
        t = t - 1.0f;
        t = t / 0.002f;
        const float sgn = (value > 0.0f) ? 1.0f : (value < 0.0f ? -1.0f : 0.0f);
        j.out_scalar[b] = sgn * (t * t - 1.0f);
    }
}

}  // namespace

void lz_launch_dense(const lz_dense_args &a, hipStream_t s)
{
    size_t lds = 0;
    for (int i = 0; i < a.njobs; ++i) {
        const int Kp = (a.job[i].K1 + a.job[i].K2 + 15) & ~15, Np = (a.job[i].N + 15) & ~15;
        const size_t need = (size_t)(16 * (Kp + 4) + 16 * (Np + 4)) * 4;
        if (need > lds) lds = need;
    }
    hipLaunchKernelGGL(k_dense, dim3((a.B + 15) / 16, a.njobs), dim3(256), lds, s, a);
}
