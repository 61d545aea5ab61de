// lz_dense.hip -- dense layers of the vector-observation (MLP) models on the fp32 matrix cores.
//
// Reference layers replaced: MLP_V2 (lzero/model/common.py:28-98) and ding.torch_utils.MLP stacks inside
// RepresentationNetworkMLP (common.py:790-851), PredictionNetworkMLP (common.py:1218-1295 and
// sampled_efficientzero_model_mlp.py), DynamicsNetwork (muzero_model_mlp.py:340-442), DynamicsNetworkMLP
// (efficientzero_model_mlp.py) and ReparameterizationHead, with eval-mode BatchNorm1d / LayerNorm, ReLU / GELU(tanh),
// the residual connection of the dynamics network and InverseScalarTransform (scaling_transform.py:82-92) in the epilogue.
#include <algorithm>

#include "lz_nn_kernels.h"
#include "lz_hinv.h"
#include "lz_wave.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// reductions on the DPP path (lz_wave.h) instead of ds_bpermute round trips: these kernels are a few thousand cycles long
__device__ __forceinline__ float red_sum(float v) { return wave_sum(v); }    // over the wave that owns the row
__device__ __forceinline__ float red_max(float v) { return wave_max(v); }
__device__ __forceinline__ float red16_sum(float v) { return group_sum<16>(v); }  // the 16 lanes (one DPP row) that share a row while staging
// max | min over the same 16 lanes (quad permutes, then the two row mirrors: every lane of the row gets the result)
template <bool MAX>
__device__ __forceinline__ float red16_ext(float v)
{
#define LZ_DPPE(ctrl) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, 0xf, 0xf, false))
    { const float o = LZ_DPPE(0xB1); v = MAX ? fmaxf(v, o) : fminf(v, o); }
    { const float o = LZ_DPPE(0x4E); v = MAX ? fmaxf(v, o) : fminf(v, o); }
    { const float o = LZ_DPPE(0x141); v = MAX ? fmaxf(v, o) : fminf(v, o); }
    { const float o = LZ_DPPE(0x140); v = MAX ? fmaxf(v, o) : fminf(v, o); }
#undef LZ_DPPE
    return v;
}
// GELU(approximate='tanh') with tanh(y) = 1 - 2 / (1 + e^{2y}) on the hardware exp / rcp (a libm tanhf is ~60 instructions
// with range branches; these kernels are a few thousand instructions in total).  |error| < 3e-7 absolute.
__device__ __forceinline__ float gelu_tanh(float u)
{
    const float y = 0.7978845608028654f * (u + 0.044715f * u * u * u);
    const float t = 1.0f - 2.0f * __frcp_rn(1.0f + __expf(2.0f * y));
    return 0.5f * u * (1.0f + t);
}
// branch-free: relu / gelu / identity selected with v_cndmask
__device__ __forceinline__ float act_fn(float u, int act)
{
    const float r = fmaxf(u, 0.0f), g = gelu_tanh(u);
    return act == 1 ? r : (act == 2 ? g : u);
}

constexpr int KCH = 16;  // k-blocks (of 16) per weight chunk held in registers

// grid = (ceil(B/16), ceil(Np/64), njobs), block = 256: wave w owns the 16-column tile 4 * blockIdx.y + w.
// MAXV = float4 per lane while staging a row (16 lanes per row): 4 covers K1 <= 256, 10 covers K1 <= 640.
// XF = some job of the launch carries a deferred transform (otherwise the gamma / beta / residual loads are not even issued).
template <int MAXV, bool XF>
__global__ __launch_bounds__(256) void k_dense(lz_dense_args a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const lz_dense_job &j = a.job[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r0 = blockIdx.x * 16, B = a.B;
    const int K1 = j.K1, K = j.K1 + j.K2, Kp = (K + 15) & ~15, PS = Kp + 4, KB = Kp >> 4;
    const int N = j.N, NT = (N + 15) >> 4;
    if ((int)blockIdx.y * 4 >= NT) return;  // jobs of one launch differ in width
    const int ct = blockIdx.y * 4 + wv;     // this wave's column tile (may be past the end: the wave then only helps staging)
    float *sX = smem;                       // [16][PS]
    // ---- first weight chunk: in flight while the input rows are staged
    const f32x4 *wp = reinterpret_cast<const f32x4 *>(j.wf) + (size_t)min(ct, NT - 1) * KB * 64 + lane;
    f32x4 bf[KCH];
#pragma unroll
    for (int q = 0; q < KCH; ++q) bf[q] = wp[(size_t)min(q, KB - 1) * 64];
    // ---- the epilogue's operands do not depend on the GEMM: requested now (unconditional; dummies where a pointer is null) --
    // loaded after the K loop they are one more exposed memory round trip per launch
    const int n_ep = min(min(ct, NT - 1) * 16 + (lane & 15), N - 1);
    const float bias_ep = j.bias[n_ep];
    const float *scp = j.scale ? j.scale : j.bias, *shp = j.scale ? j.shift : j.bias;
    const float sc_ep = scp[n_ep], sh_ep = shp[n_ep];
    // ---- stage 16 input rows: 16 lanes per row (same wavefront), producer's LayerNorm / activation / residual on the fly.
    // Every global load of the pass is issued before anything is consumed (straight-line, float4 granularity): a loop of
    // load -> use iterations would pay one memory round trip per element.
    {
        const int row = tid >> 4, part = tid & 15;
        const int b = min(r0 + row, B - 1);
        const float *src = j.x + (j.x_gather ? (size_t)j.x_gather[b] * (size_t)j.x_slot_stride : 0) + (size_t)b * K1;
        float *dst = sX + row * PS;
        // second input (the action encoding, K2 <= 16 in every model): its value for this lane's first tail column is requested
        // with the row, not after it
        float x2_first = 0.0f;
        if (j.K2 > 0) {
            const int kk0 = min(part, j.K2 - 1);
            if (j.x2_mode == 1) x2_first = j.x2[(size_t)b * j.K2 + kk0];
            else x2_first = (float)j.x2_idx[b];
        }
        if ((K1 & 3) == 0) {
            // unconditional, clamped loads (a predicated load makes the compiler branch around it and drain the load queue);
            // absent transforms read the row itself as a dummy and are switched off with selects
            const int nv = K1 >> 2;
            const bool has_ln = j.in_ln_g != nullptr, has_res = j.in_res != nullptr;
            const float *gp = has_ln ? j.in_ln_g : src, *bp = has_ln ? j.in_ln_b : src;
            const float *rp = has_res ? j.in_res + (j.in_res_gather ? (size_t)j.in_res_gather[b] * (size_t)j.in_res_slot_stride : 0) + (size_t)b * K1 : src;
            f32x4 xv[MAXV], gv[XF ? MAXV : 1], bv[XF ? MAXV : 1], rv[XF ? MAXV : 1];
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int ic = 4 * min(part + 16 * i, nv - 1);
                xv[i] = *reinterpret_cast<const f32x4 *>(src + ic);
                if constexpr (XF) {
                    gv[i] = *reinterpret_cast<const f32x4 *>(gp + ic);
                    bv[i] = *reinterpret_cast<const f32x4 *>(bp + ic);
                    rv[i] = *reinterpret_cast<const f32x4 *>(rp + ic);
                }
            }
            if constexpr (!XF) {
#pragma unroll
                for (int i = 0; i < MAXV; ++i) {
                    const int idx = part + 16 * i;
                    if (idx < nv) *reinterpret_cast<f32x4 *>(dst + 4 * idx) = xv[i];
                }
            } else {
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const float s4 = (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
                sum += (part + 16 * i < nv) ? s4 : 0.0f;
            }
            const float mean_ln = red16_sum(sum) / (float)K1;
            float sq = 0.0f;
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                float q4 = 0.0f;
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float d = xv[i][q] - mean_ln; q4 += d * d; }
                sq += (part + 16 * i < nv) ? q4 : 0.0f;
            }
            const float rstd_ln = 1.0f / sqrtf(red16_sum(sq) / (float)K1 + j.in_ln_eps);
            const float mean = has_ln ? mean_ln : 0.0f, rstd = has_ln ? rstd_ln : 1.0f;
            const bool wr = j.in_out && blockIdx.y == 0 && r0 + row < B;
            // the transformed row stays in xv; state_norm=True (in_minmax) then renormalises it over the row: (x - min) / max(max - min, 1e-8),
            // the same fp32 subtraction and division as lzero/model/utils.py:261-269
            float rmn = __builtin_inff(), rmx = -__builtin_inff();
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float u = (xv[i][q] - mean) * rstd;
                    u = has_ln ? u * gv[i][q] + bv[i][q] : u;
                    u = act_fn(u, j.in_act);
                    u += has_res ? rv[i][q] : 0.0f;
                    xv[i][q] = u;
                    if (part + 16 * i < nv) { rmn = fminf(rmn, u); rmx = fmaxf(rmx, u); }
                }
            }
            if (j.in_minmax != 0) {   // (uniform over the workgroup: a scalar branch around the divisions)
                rmn = red16_ext<false>(rmn); rmx = red16_ext<true>(rmx);
                const float d = rmx - rmn;
                const float mm_den = d < 1e-8f ? 1e-8f : d;
#pragma unroll
                for (int i = 0; i < MAXV; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) xv[i][q] = (xv[i][q] - rmn) / mm_den;
            }
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int idx = part + 16 * i;
                const f32x4 v = xv[i];
                if (idx < nv) {
                    *reinterpret_cast<f32x4 *>(dst + 4 * idx) = v;
                    if (wr) *reinterpret_cast<f32x4 *>(j.in_out + (size_t)(r0 + row) * K1 + 4 * idx) = v;
                }
            }
            }
        } else {  // tiny odd widths (raw observations): no transform is ever attached to them
            for (int k = part; k < K1; k += 16) dst[k] = src[k];
        }
        for (int k = K1 + part; k < Kp; k += 16) {
            float v = 0.0f;
            if (k < K) {
                const int kk = k - K1;
                if (kk == part && part < j.K2) {  // first tail column of this lane: preloaded
                    if (j.x2_mode == 1) v = x2_first;
                    else if (j.x2_mode == 2) v = ((int)x2_first == kk) ? 1.0f : 0.0f;
                    else v = x2_first / j.x2_div;
                } else if (j.x2_mode == 1) v = j.x2[(size_t)b * j.K2 + kk];
                else if (j.x2_mode == 2) v = (j.x2_idx[b] == kk) ? 1.0f : 0.0f;
                else v = (float)j.x2_idx[b] / j.x2_div;
            }
            dst[k] = v;
        }
    }
    __syncthreads();
    if (ct >= NT) return;
    // ---- GEMM: one 16 x 16 tile per wave
    const float *sAf = sX + (lane & 15) * PS + (lane >> 4) * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < KB; k0 += KCH) {
        if (k0 > 0) {
#pragma unroll
            for (int q = 0; q < KCH; ++q) bf[q] = wp[(size_t)min(k0 + q, KB - 1) * 64];
        }
#pragma unroll
        for (int q = 0; q < KCH; ++q) {
            if (k0 + q < KB) {
                const f32x4 af = *reinterpret_cast<const f32x4 *>(sAf + (k0 + q) * 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[q][i], acc, 0, 0, 0);
            }
        }
    }
    // ---- epilogue: bias (+ BN) (+ activation when nothing is deferred) (+ the reparameterisation head's transforms)
    const int n = ct * 16 + (lane & 15);
    if (n >= N) return;
    const float bias = bias_ep;
    const float sc = j.scale ? sc_ep : 1.0f, sh = j.scale ? sh_ep : 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int b = r0 + 4 * (lane >> 4) + q;
        if (b >= B) continue;
        float u = act_fn((acc[q] + bias) * sc + sh, j.act);
        if (j.final == 2) {
            if (n >= j.final_split) u = expf(fminf(fmaxf(u, -20.0f), 2.0f));
            else if (j.final_tanh) u = tanhf(u);
        }
        j.out[(size_t)b * N + n] = u;
    }
}

// The same layer for WIDE raw inputs (a representation network's first Linear on a long observation vector: MiniGrid's 2835 features,
// zoo/minigrid/config/minigrid_*_config.py): the 16 input rows pass through LDS in chunks of 512 columns, the accumulator tile stays in
// registers across the chunks.  No deferred input transform, no second input block (raw observations have neither).
// grid = (ceil(B/16), ceil(N/64)), block = 256: wave w owns the 16-column tile 4 * blockIdx.y + w.
constexpr int WCH = 512;   // input columns per chunk
__global__ __launch_bounds__(256) void k_dense_wide(lz_dense_args a)
{
    __shared__ __attribute__((aligned(16))) float sX[16 * (WCH + 4)];
    const lz_dense_job &j = a.job[0];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r0 = blockIdx.x * 16, B = a.B, K = j.K1, Kp = (K + 15) & ~15, KB = Kp >> 4, PS = WCH + 4;
    const int N = j.N, NT = (N + 15) >> 4, ct = blockIdx.y * 4 + wv;
    const f32x4 *wp = reinterpret_cast<const f32x4 *>(j.wf) + (size_t)min(ct, NT - 1) * KB * 64 + lane;
    const int n_ep = min(min(ct, NT - 1) * 16 + (lane & 15), N - 1);
    const float bias_ep = j.bias[n_ep];
    const float *scp = j.scale ? j.scale : j.bias, *shp = j.scale ? j.shift : j.bias;
    const float sc_ep = scp[n_ep], sh_ep = shp[n_ep];
    const float *sAf = sX + (lane & 15) * PS + (lane >> 4) * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < Kp; c0 += WCH) {
        const int cw = min(WCH, Kp - c0);            // columns of this chunk (a multiple of 16)
        if (c0 > 0) __syncthreads();                 // every wave is done with the previous chunk
        {
            const int row = tid >> 4, part = tid & 15;
            const float *src = j.x + (size_t)min(r0 + row, B - 1) * K;
            for (int k = part; k < cw; k += 16) sX[row * PS + k] = (c0 + k < K) ? src[c0 + k] : 0.0f;
        }
        __syncthreads();
        if (ct < NT) {
            const int kb0 = c0 >> 4, nkb = cw >> 4;
            for (int q0 = 0; q0 < nkb; q0 += 8) {
                f32x4 bf[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) bf[q] = wp[(size_t)min(kb0 + q0 + q, KB - 1) * 64];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (q0 + q < nkb) {
                        const f32x4 af = *reinterpret_cast<const f32x4 *>(sAf + (q0 + q) * 16);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[q][i], acc, 0, 0, 0);
                    }
                }
            }
        }
    }
    const int n = ct * 16 + (lane & 15);
    if (ct >= NT || n >= N) return;
    const float sc = j.scale ? sc_ep : 1.0f, sh = j.scale ? sh_ep : 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int b = r0 + 4 * (lane >> 4) + q;
        if (b < B) j.out[(size_t)b * N + n] = act_fn((acc[q] + bias_ep) * sc + sh, j.act);
    }
}

// grid = (ceil(B/4), njobs), block = 256: one wave per row
__global__ __launch_bounds__(256) void k_rowfinal(lz_rowfinal_args a)
{
    const lz_rowfinal_job &j = a.job[blockIdx.y];
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= a.B) return;
    const int N = j.N;
    const float *y = j.logits + (size_t)b * N;
    if (j.scalar) {   // categorical_distribution=False: value = logits (scaling_transform.py:88-89), then the same h^-1
        if (lane == 0) j.out_scalar[b] = lz_inverse_scalar_transform(y[0]);
        return;
    }
    float mx = -__builtin_inff();
    for (int c = lane; c < N; c += 64) mx = fmaxf(mx, y[c]);
    mx = red_max(mx);
    float s0 = 0.0f, s1 = 0.0f;
    for (int c = lane; c < N; c += 64) {
        const float ex = expf(y[c] - mx);
        s0 += ex;
        s1 += ex * (j.support_min + (float)c);
    }
    s0 = red_sum(s0);
    s1 = red_sum(s1);
    if (lane == 0) {
        // softmax . support, then InverseScalarTransform.__call__ (scaling_transform.py:82-92) in torch's fp32 op order (lz_hinv.h)
        const float value = s1 / s0;
        j.out_scalar[b] = lz_inverse_scalar_transform(value);
    }
}

__global__ __launch_bounds__(256) void k_hinv_dense(const float *__restrict__ in, float *__restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = lz_inverse_scalar_transform(in[i]);
}

}  // namespace

void lz_launch_hinv_dense(const float *d_in, float *d_out, int64_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_hinv_dense, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_in, d_out, n);
}

void lz_launch_dense(const lz_dense_args &a, hipStream_t s)
{
    if (a.njobs == 1 && a.job[0].K1 > 640) {   // a wide raw input (lz_model_finalize admits it for the first representation layer only)
        hipLaunchKernelGGL(k_dense_wide, dim3((a.B + 15) / 16, (a.job[0].N + 63) / 64), dim3(256), 0, s, a);
        return;
    }
    size_t lds = 0;
    int ncg = 1;
    for (int i = 0; i < a.njobs; ++i) {
        const int Kp = (a.job[i].K1 + a.job[i].K2 + 15) & ~15, cg = (a.job[i].N + 63) / 64;
        lds = std::max(lds, (size_t)16 * (Kp + 4) * 4);
        ncg = std::max(ncg, cg);
    }
    int k1max = 0;
    for (int i = 0; i < a.njobs; ++i) k1max = std::max(k1max, a.job[i].K1);
    bool xf = false;
    for (int i = 0; i < a.njobs; ++i) xf = xf || a.job[i].in_ln_g || a.job[i].in_act || a.job[i].in_res || a.job[i].in_out || a.job[i].in_minmax;
    const dim3 grid((a.B + 15) / 16, ncg, a.njobs), block(256);
    if (k1max <= 256) {
        if (xf) hipLaunchKernelGGL((k_dense<4, true>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((k_dense<4, false>), grid, block, lds, s, a);
    } else if (k1max <= 640) {
        if (xf) hipLaunchKernelGGL((k_dense<10, true>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((k_dense<10, false>), grid, block, lds, s, a);
    } else {   // up to 1024 columns: the first head layers of the conv Sampled EfficientZero on an 8x8 latent (16 x 64)
        if (xf) hipLaunchKernelGGL((k_dense<16, true>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((k_dense<16, false>), grid, block, lds, s, a);
    }
}

void lz_launch_rowfinal(const lz_rowfinal_args &a, hipStream_t s)
{
    hipLaunchKernelGGL(k_rowfinal, dim3((a.B + 3) / 4, a.njobs), dim3(256), 0, s, a);
}
