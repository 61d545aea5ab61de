// lz_nn.hip -- fp32 network kernels for gfx950 (MI355X): implicit-GEMM 3x3 convolution and LSTM gate
// GEMM on the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32), heads / pooling on the vector ALU.
//
// What they compute (reference = LightZero v0.2.0, evaluated in eval() mode):
//   conv3x3 + folded BatchNorm (+ residual) (+ ReLU)   ding ResBlock as used by lzero/model/common.py:266-365,
//                                                      :706-787, :1081-1216 and efficientzero_model.py:545-554
//   one-hot action planes of the dynamics conv          efficientzero_model.py:335-373 (as a per-action table)
//   LSTM step + BatchNorm1d + ReLU                      efficientzero_model.py:559-566
//   conv1x1 + BN + ReLU -> MLP heads                    common.py:1196-1216, efficientzero_model.py:556-567
//   softmax . support -> h^-1                           lzero/policy/scaling_transform.py:82-92
//
// Tiling rationale (MI355X: 256 CUs x 4 SIMDs, fp32 MFMA 16x16x4 = 32 cycles/SIMD): at 256 roots a 6x6x64
// layer is only 9216 x 64 outputs = 2304 MFMA tiles = exactly 9 per CU.  A workgroup therefore owns
// 144 output pixels x 16 output channels (9 tiles), its four waves split the reduction (K) dimension so
// that every SIMD of every CU issues the same 324 MFMAs, and the partial tiles meet in LDS.  The input
// halo and the 16-channel weight slice are staged in LDS once per workgroup with a +4-float pixel
// pad, which makes every ds_read_b128 of an operand fragment bank-conflict free.
#include "lz_nn_kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

template <int VEC> struct vecf;
template <> struct vecf<4> { typedef float4 type; };
template <> struct vecf<2> { typedef float2 type; };

__device__ __forceinline__ float vget(const float4 &v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }
__device__ __forceinline__ float vget(const float2 &v, int j) { return j == 0 ? v.x : v.y; }
__device__ __forceinline__ float4 vzero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// ------------------------------------------------------------------------------------------------
// 3x3 convolution as implicit GEMM.  grid = (ceil(B*Hout*Wout / 144), Cout / 16), block = 256.
// ------------------------------------------------------------------------------------------------
template <int CIN, int STRIDE>
__global__ __launch_bounds__(256) void k_conv3x3(lz_conv_args a, int npix_max)
{
    constexpr int VEC = CIN / 16;   // floats per lane per operand fetch (64 -> b128, 32 -> b64)
    constexpr int PS = CIN + 4;     // padded pixel stride (floats): conflict-free fragment reads
    constexpr int TM = 144, MT = 9;
    typedef typename vecf<VEC>::type vec_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sA = smem;
    float *sB = smem + (size_t)npix_max * PS;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int HWin = a.Hin * a.Win, HWout = a.Hout * a.Wout;
    const int M = a.B * HWout;
    const int m0 = blockIdx.x * TM;
    const int m1 = min(m0 + TM, M) - 1;
    const int n0 = blockIdx.y * 16;

    auto centre = [&](int m) -> int {  // flat NHWC pixel index of the input under the centre tap
        if (STRIDE == 1) return m;
        const int b = m / HWout, p = m - b * HWout, y = p / a.Wout, x = p - y * a.Wout;
        return (b * a.Hin + STRIDE * y) * a.Win + STRIDE * x;
    };
    const int in_lo = max(0, centre(m0) - a.Win - 1);
    const int in_hi = min(a.B * HWin, centre(m1) + a.Win + 2);
    const int npix = in_hi - in_lo;

    // ---- stage the input halo (a contiguous pixel range of the NHWC tensor) and the weight slice
    constexpr int CH4 = CIN / 4;
    for (int idx = tid; idx < npix * CH4; idx += 256) {
        const int pix = idx / CH4, c4 = idx - pix * CH4;
        const int q = in_lo + pix;
        const float *src;
        if (a.gather_ix) {
            const int b = q / HWin;
            src = a.in + (size_t)a.gather_ix[b] * a.slot_stride + (size_t)q * CIN;
        } else {
            src = a.in + (size_t)q * CIN;
        }
        *reinterpret_cast<float4 *>(sA + (size_t)pix * PS + c4 * 4) = *reinterpret_cast<const float4 *>(src + c4 * 4);
    }
    {
        const float *wsrc = a.w + (size_t)blockIdx.y * 9 * 16 * CIN;
        for (int idx = tid; idx < 9 * 16 * CH4; idx += 256) {
            const int row = idx / CH4, c4 = idx - row * CH4;
            *reinterpret_cast<float4 *>(sB + (size_t)row * PS + c4 * 4) =
                *reinterpret_cast<const float4 *>(wsrc + (size_t)row * CIN + c4 * 4);
        }
    }
    // ---- per-lane geometry of its row in each of the 9 M-tiles
    int base[MT], mask[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + i * 16 + (lane & 15);
        int mk = 0, bs = 0;
        if (m <= m1) {
            const int b = m / HWout, p = m - b * HWout, y = p / a.Wout, x = p - y * a.Wout;
            const int cy = STRIDE * y, cx = STRIDE * x;
            bs = ((b * a.Hin + cy) * a.Win + cx - in_lo) * PS;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = cy + t / 3 - 1, ix = cx + t % 3 - 1;
                if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) mk |= 1 << t;
            }
        }
        base[i] = bs;
        mask[i] = mk;
    }
    __syncthreads();

    // ---- K loop: wave wv owns channel group wv (16 or 8 channels) of every tap
    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int coff = wv * 4 * VEC + (lane >> 4) * VEC;
    const float *sBl = sB + (size_t)(lane & 15) * PS + coff;
    const float *sAl = sA + coff;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int toff = ((t / 3 - 1) * a.Win + (t % 3 - 1)) * PS;
        const vec_t bf = *reinterpret_cast<const vec_t *>(sBl + (size_t)t * 16 * PS);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const bool v = (mask[i] >> t) & 1;
            const int off = v ? base[i] + toff : 0;
            vec_t af = *reinterpret_cast<const vec_t *>(sAl + off);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float av = v ? vget(af, j) : 0.0f;
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, vget(bf, j), acc[i], 0, 0, 0);
            }
        }
    }
    // ---- cross-wave (split-K) reduction through LDS, fused epilogue
    __syncthreads();
    float *red = smem;  // [4][MT][4][64]
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wv * MT + i) * 4 + r) * 64 + lane] = acc[i][r];
    __syncthreads();
    const int l = tid & 63, r = tid >> 6;
    const int col = l & 15, row_in_tile = 4 * (l >> 4) + r;  // C/D layout of mfma_f32_16x16x4
    const int co = n0 + col;
    const float sc = a.scale[co], sh = a.shift[co];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + i * 16 + row_in_tile;
        if (m > m1) continue;
        float v = red[((0 * MT + i) * 4 + r) * 64 + l] + red[((1 * MT + i) * 4 + r) * 64 + l] +
                  red[((2 * MT + i) * 4 + r) * 64 + l] + red[((3 * MT + i) * 4 + r) * 64 + l];
        const int b = m / HWout, p = m - b * HWout;
        if (a.act_table) v += a.act_table[((size_t)a.action[b] * HWout + p) * a.Cout + co];
        v = v * sc + sh;
        if (a.residual) {
            const float *rp = a.residual;
            if (a.residual_gather) rp += (size_t)a.gather_ix[b] * a.slot_stride;
            v += rp[(size_t)m * a.Cout + co];
        }
        if (a.relu) v = fmaxf(v, 0.0f);
        a.out[(size_t)m * a.Cout + co] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// first DownSample layer: conv3x3 / stride 2 from NCHW observations, + BN + ReLU.  One thread per
// output pixel computes all Cout channels from the (<= 9*C)-value patch; weights broadcast from LDS.
// ------------------------------------------------------------------------------------------------
template <int C, int COUT>
__global__ __launch_bounds__(256) void k_conv_first(const float *__restrict__ obs, const float *__restrict__ w,
                                                    const float *__restrict__ scale, const float *__restrict__ shift,
                                                    float *__restrict__ out, int B, int H, int W)
{
    // 64 output pixels per block; thread = (pixel, group of 8 output channels)
    constexpr int G = COUT / 8;
    static_assert(G == 4, "block layout assumes 4 channel groups");
    __shared__ float sw[9 * C * COUT];
    for (int i = threadIdx.x; i < 9 * C * COUT; i += 256) sw[i] = w[i];
    __syncthreads();
    const int Ho = H / 2, Wo = W / 2;
    const int g = threadIdx.x & 3;
    const int64_t m = (int64_t)blockIdx.x * 64 + (threadIdx.x >> 2);
    if (m >= (int64_t)B * Ho * Wo) return;
    const int b = (int)(m / (Ho * Wo)), p = (int)(m - (int64_t)b * Ho * Wo), y = p / Wo, x = p - y * Wo;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.0f;
#pragma unroll 1
    for (int t = 0; t < 9; ++t) {
        const int iy = 2 * y + t / 3 - 1, ix = 2 * x + t % 3 - 1;
        const bool v = iy >= 0 && iy < H && ix >= 0 && ix < W;
#pragma unroll
        for (int ci = 0; ci < C; ++ci) {
            const float xv = v ? obs[(((size_t)b * C + ci) * H + iy) * W + ix] : 0.0f;
            const float *wr = sw + (t * C + ci) * COUT + g * 8;
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] += xv * wr[c];
        }
    }
    float *o = out + (size_t)m * COUT + g * 8;
    const float *sc = scale + g * 8, *sh = shift + g * 8;
#pragma unroll
    for (int c = 0; c < 8; c += 4) {
        float4 v;
        v.x = fmaxf(acc[c + 0] * sc[c + 0] + sh[c + 0], 0.f);
        v.y = fmaxf(acc[c + 1] * sc[c + 1] + sh[c + 1], 0.f);
        v.z = fmaxf(acc[c + 2] * sc[c + 2] + sh[c + 2], 0.f);
        v.w = fmaxf(acc[c + 3] * sc[c + 3] + sh[c + 3], 0.f);
        *reinterpret_cast<float4 *>(o + c) = v;
    }
}

// AvgPool2d(3, stride 2, pad 1), count_include_pad=True (divide by 9), NHWC, float4 per thread
__global__ __launch_bounds__(256) void k_avgpool(const float *__restrict__ in, float *__restrict__ out, int B, int Hin,
                                                 int Win, int C)
{
    const int Ho = (Hin + 1) / 2, Wo = (Win + 1) / 2, C4 = C / 4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)B * Ho * Wo * C4) return;
    const int c4 = (int)(idx % C4);
    const int64_t pix = idx / C4;
    const int x = (int)(pix % Wo), y = (int)((pix / Wo) % Ho), b = (int)(pix / ((int64_t)Wo * Ho));
    float4 s = vzero4();
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int iy = 2 * y + t / 3 - 1, ix = 2 * x + t % 3 - 1;
        if (iy >= 0 && iy < Hin && ix >= 0 && ix < Win) {
            const float4 v = *reinterpret_cast<const float4 *>(in + (((size_t)b * Hin + iy) * Win + ix) * C + c4 * 4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    s.x /= 9.0f; s.y /= 9.0f; s.z /= 9.0f; s.w /= 9.0f;
    *reinterpret_cast<float4 *>(out + (size_t)pix * C + c4 * 4) = s;
}

// conv1x1 + bias + BN + ReLU, NHWC; block = 256 threads handles 64 pixels x Cout (<= 32) outputs
template <int CIN>
__global__ __launch_bounds__(256) void k_conv1x1(const float *__restrict__ in, const float *__restrict__ w,
                                                 const float *__restrict__ bias, const float *__restrict__ scale,
                                                 const float *__restrict__ shift, float *__restrict__ out, int npix,
                                                 int Cout)
{
    __shared__ float sw[32 * (CIN + 1)];
    __shared__ float sx[64 * (CIN + 1)];
    for (int i = threadIdx.x; i < Cout * CIN; i += 256) sw[(i / CIN) * (CIN + 1) + i % CIN] = w[i];
    const int p0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * CIN; i += 256) {
        const int p = p0 + i / CIN;
        sx[(i / CIN) * (CIN + 1) + i % CIN] = p < npix ? in[(size_t)p * CIN + i % CIN] : 0.0f;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 64 * Cout; o += 256) {
        const int pl = o / Cout, co = o - pl * Cout;
        if (p0 + pl >= npix) continue;
        float acc = 0.0f;
#pragma unroll 8
        for (int c = 0; c < CIN; ++c) acc += sx[pl * (CIN + 1) + c] * sw[co * (CIN + 1) + c];
        acc += bias[co];
        acc = acc * scale[co] + shift[co];
        out[(size_t)(p0 + pl) * Cout + co] = fmaxf(acc, 0.0f);
    }
}

// ------------------------------------------------------------------------------------------------
// LSTM step.  gates[B][4H] = [x | h] . Wcat^T ; tile = 64 rows x 32 gate columns (= 8 hidden units),
// K streamed through LDS in 64-wide chunks (double buffered), 4 waves split each chunk's K.
// grid = (ceil(B/64), 4H/32), block = 256.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void k_lstm(lz_lstm_args a)
{
    constexpr int KC = 64, PS = KC + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    auto sA = [&](int buf) -> float * { return smem + buf * 64 * PS; };
    auto sB = [&](int buf) -> float * { return smem + 2 * 64 * PS + buf * 32 * PS; };
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r0 = blockIdx.x * 64, n0 = blockIdx.y * 32;
    const int K = a.KX + a.H, nchunk = K / KC;
    const size_t slot = (size_t)a.B * a.H;

    float4 ra[4], rb[2];
    auto load_chunk = [&](int c) {
        const int k0 = c * KC;
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // A chunk: 64 rows x 16 float4
            const int idx = tid + i * 256, row = idx >> 4, c4 = idx & 15;
            const int b = r0 + row;
            float4 v = vzero4();
            if (b < a.B) {
                const float *src = (k0 < a.KX) ? a.x + (size_t)b * a.KX + k0
                                               : a.h_pool + (size_t)a.gather_ix[b] * slot + (size_t)b * a.H + (k0 - a.KX);
                v = *reinterpret_cast<const float4 *>(src + c4 * 4);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {  // B chunk: 32 gate rows x 16 float4
            const int idx = tid + i * 256, row = idx >> 4, c4 = idx & 15;
            rb[i] = *reinterpret_cast<const float4 *>(a.wcat + (size_t)(n0 + row) * K + k0 + c4 * 4);
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * 256, row = idx >> 4, c4 = idx & 15;
            *reinterpret_cast<float4 *>(sA(buf) + row * PS + c4 * 4) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * 256, row = idx >> 4, c4 = idx & 15;
            *reinterpret_cast<float4 *>(sB(buf) + row * PS + c4 * 4) = rb[i];
        }
    };

    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int coff = wv * 16 + (lane >> 4) * 4;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) load_chunk(c + 1);
        float4 bf[2], af[4];
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const float4 *>(sB(buf) + (j * 16 + (lane & 15)) * PS + coff);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const float4 *>(sA(buf) + (i * 16 + (lane & 15)) * PS + coff);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(vget(af[i], q), vget(bf[j], q), acc[i][j], 0, 0, 0);
        if (c + 1 < nchunk) store_chunk(buf ^ 1);
        __syncthreads();
    }
    // ---- split-K reduction, then the LSTM cell for 64 rows x 8 units
    float *red = smem;                       // [4][8][4][64]   (32 KB)
    float *csum = smem + 4 * 8 * 4 * 64;     // [64][33]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((wv * 8 + i * 2 + j) * 4 + r) * 64 + lane] = acc[i][j][r];
    __syncthreads();
    {
        const int l = tid & 63, r = tid >> 6;
#pragma unroll
        for (int tile = 0; tile < 8; ++tile) {
            const float v = red[((0 * 8 + tile) * 4 + r) * 64 + l] + red[((1 * 8 + tile) * 4 + r) * 64 + l] +
                            red[((2 * 8 + tile) * 4 + r) * 64 + l] + red[((3 * 8 + tile) * 4 + r) * 64 + l];
            const int row = (tile >> 1) * 16 + 4 * (l >> 4) + r, col = (tile & 1) * 16 + (l & 15);
            csum[row * 33 + col] = v;
        }
    }
    __syncthreads();
    for (int item = tid; item < 64 * 8; item += 256) {
        const int row = item >> 3, u = item & 7;
        const int b = r0 + row;
        if (b >= a.B) continue;
        const int unit = (n0 >> 2) + u;
        const float gi = csum[row * 33 + 4 * u + 0] + a.bias[n0 + 4 * u + 0];
        const float gf = csum[row * 33 + 4 * u + 1] + a.bias[n0 + 4 * u + 1];
        const float gg = csum[row * 33 + 4 * u + 2] + a.bias[n0 + 4 * u + 2];
        const float go = csum[row * 33 + 4 * u + 3] + a.bias[n0 + 4 * u + 3];
        const float c_prev = a.c_pool[(size_t)a.gather_ix[b] * slot + (size_t)b * a.H + unit];
        const float cn = sigmoidf_(gf) * c_prev + sigmoidf_(gi) * tanhf(gg);
        const float hn = sigmoidf_(go) * tanhf(cn);
        bool reset = false;
        if (a.search_len && a.horizon > 0) reset = (a.search_len[b] % a.horizon) == 0;  // mcts_ctree.py:859-863
        a.h_out[(size_t)b * a.H + unit] = reset ? 0.0f : hn;
        a.c_out[(size_t)b * a.H + unit] = reset ? 0.0f : cn;
        a.hbn_out[(size_t)b * a.H + unit] = fmaxf(hn * a.bn_scale[unit] + a.bn_shift[unit], 0.0f);
    }
}

// ------------------------------------------------------------------------------------------------
// heads.  grid = (ceil(B/4), nheads), block = 256: four roots share every weight fetch.
// ------------------------------------------------------------------------------------------------
constexpr int EPB = 4;
constexpr int MAXH = 4;
struct head_pack { lz_head_desc h[MAXH]; };

__device__ __forceinline__ float block_reduce(float v, float *scratch, bool is_max)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float t = __shfl_xor(v, o);
        v = is_max ? fmaxf(v, t) : v + t;
    }
    const int wv = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[wv] = v;
    __syncthreads();
    const float a0 = scratch[0], a1 = scratch[1], a2 = scratch[2], a3 = scratch[3];
    return is_max ? fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)) : (a0 + a1) + (a2 + a3);
}

__global__ __launch_bounds__(256) void k_heads(head_pack hp, int B, int HW, int C, int HC, int HID)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const lz_head_desc &h = hp.h[blockIdx.y];
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * EPB;
    const int K1 = h.K1;
    float *xs = smem;                    // [EPB][K1]
    float *hid = xs + EPB * K1;          // [EPB][HID]
    float *scr = hid + EPB * HID;        // [8]
    float *lat = scr + 8;                // [EPB][HW][C+1] (conv heads only)
    if (h.has_conv) {
        const int CP = C + 1;
        for (int i = tid; i < EPB * HW * C; i += 256) {
            const int e = i / (HW * C), rem = i - e * HW * C;
            const int b = b0 + e;
            lat[(e * HW + rem / C) * CP + rem % C] = b < B ? h.in[(size_t)b * HW * C + rem] : 0.0f;
        }
        __syncthreads();
        for (int o = tid; o < EPB * HW * HC; o += 256) {
            const int e = o / (HW * HC), rem = o - e * HW * HC, p = rem / HC, co = rem - p * HC;
            const float *lp = lat + (e * HW + p) * CP;
            const float *wr = h.cw + (size_t)co * C;
            float acc = 0.0f;
            for (int c = 0; c < C; ++c) acc += lp[c] * wr[c];
            acc += h.cb[co];
            acc = acc * h.cscale[co] + h.cshift[co];
            xs[e * K1 + rem] = fmaxf(acc, 0.0f);  // K1 index = pixel*HC + channel
        }
    } else {
        for (int i = tid; i < EPB * K1; i += 256) {
            const int e = i / K1, b = b0 + e;
            xs[i] = b < B ? h.in[(size_t)b * K1 + (i - e * K1)] : 0.0f;
        }
    }
    __syncthreads();
    // ---- layer 1: HID(=32) units x 8 K-parts, float4 weight fetches (128 B contiguous per unit)
    {
        const int u = tid >> 3, part = tid & 7;
        float acc[EPB];
#pragma unroll
        for (int e = 0; e < EPB; ++e) acc[e] = 0.0f;
        if (u < HID) {
            const float *wr = h.w1 + (size_t)u * K1;
            for (int k = part * 4; k < K1; k += 32) {
                const float4 wv4 = *reinterpret_cast<const float4 *>(wr + k);
#pragma unroll
                for (int e = 0; e < EPB; ++e) {
                    const float4 xv = *reinterpret_cast<const float4 *>(xs + e * K1 + k);
                    acc[e] += wv4.x * xv.x + wv4.y * xv.y + wv4.z * xv.z + wv4.w * xv.w;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < EPB; ++e) {
            acc[e] += __shfl_xor(acc[e], 1);
            acc[e] += __shfl_xor(acc[e], 2);
            acc[e] += __shfl_xor(acc[e], 4);
        }
        if (u < HID && part == 0) {
#pragma unroll
            for (int e = 0; e < EPB; ++e) hid[e * HID + u] = fmaxf((acc[e] + h.b1[u]) * h.s1[u] + h.t1[u], 0.0f);
        }
    }
    __syncthreads();
    // ---- layer 2 (+ softmax expectation over the support, + h^-1)
    constexpr int NPT = 3;  // outputs per thread: NOUT <= 768
    float lg[NPT][EPB];
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int n = tid + i * 256;
#pragma unroll
        for (int e = 0; e < EPB; ++e) lg[i][e] = -__builtin_inff();
        if (n < h.NOUT) {
            const float *wr = h.w2 + (size_t)n * HID;
            float acc[EPB];
#pragma unroll
            for (int e = 0; e < EPB; ++e) acc[e] = 0.0f;
            for (int k = 0; k < HID; k += 4) {
                const float4 wv4 = *reinterpret_cast<const float4 *>(wr + k);
#pragma unroll
                for (int e = 0; e < EPB; ++e) {
                    const float4 hv = *reinterpret_cast<const float4 *>(hid + e * HID + k);
                    acc[e] += wv4.x * hv.x + wv4.y * hv.y + wv4.z * hv.z + wv4.w * hv.w;
                }
            }
#pragma unroll
            for (int e = 0; e < EPB; ++e) {
                lg[i][e] = acc[e] + h.b2[n];
                if (h.out_logits && b0 + e < B) h.out_logits[(size_t)(b0 + e) * h.NOUT + n] = lg[i][e];
            }
        }
    }
    if (!h.categorical) return;
#pragma unroll
    for (int e = 0; e < EPB; ++e) {
        float m = fmaxf(fmaxf(lg[0][e], lg[1][e]), lg[2][e]);
        m = block_reduce(m, scr, true);
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int n = tid + i * 256;
            if (n < h.NOUT) {
                const float ex = expf(lg[i][e] - m);
                s0 += ex;
                s1 += ex * (h.support_min + (float)n);
            }
        }
        s0 = block_reduce(s0, scr, false);
        s1 = block_reduce(s1, scr, false);
        if (tid == 0 && b0 + e < B) {
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
            h.out_scalar[b0 + e] = sgn * (t * t - 1.0f);
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static int conv_npix_max(const lz_conv_args &a, int stride)
{
    if (stride == 1) return 144 + 2 * a.Win + 2;
    return 2 * a.Win * (143 / a.Wout + 1) + 2 * (a.Wout - 1) + 2 * a.Win + 3;
}

void lz_launch_conv3x3(const lz_conv_args &a, int cin, int stride, hipStream_t s)
{
    const int M = a.B * a.Hout * a.Wout;
    int npix = conv_npix_max(a, stride);
    npix = min(npix, a.B * a.Hin * a.Win);
    const int ps = cin + 4;
    size_t lds = ((size_t)npix * ps + (size_t)9 * 16 * ps) * 4;
    const size_t red = (size_t)4 * 9 * 4 * 64 * 4;
    if (lds < red) lds = red;
    dim3 grid((M + 143) / 144, a.Cout / 16), block(256);
    if (cin == 64 && stride == 1) hipLaunchKernelGGL((k_conv3x3<64, 1>), grid, block, lds, s, a, npix);
    else if (cin == 32 && stride == 1) hipLaunchKernelGGL((k_conv3x3<32, 1>), grid, block, lds, s, a, npix);
    else if (cin == 32 && stride == 2) hipLaunchKernelGGL((k_conv3x3<32, 2>), grid, block, lds, s, a, npix);
    else if (cin == 64 && stride == 2) hipLaunchKernelGGL((k_conv3x3<64, 2>), grid, block, lds, s, a, npix);
}

void lz_launch_conv_first(const float *obs, const float *w, const float *scale, const float *shift, float *out, int B,
                          int C, int H, int W, int Cout, hipStream_t s)
{
    const int64_t M = (int64_t)B * (H / 2) * (W / 2);
    dim3 grid((unsigned)((M + 63) / 64)), block(256);
    if (C == 4 && Cout == 32) hipLaunchKernelGGL((k_conv_first<4, 32>), grid, block, 0, s, obs, w, scale, shift, out, B, H, W);
    else if (C == 1 && Cout == 32) hipLaunchKernelGGL((k_conv_first<1, 32>), grid, block, 0, s, obs, w, scale, shift, out, B, H, W);
    else if (C == 3 && Cout == 32) hipLaunchKernelGGL((k_conv_first<3, 32>), grid, block, 0, s, obs, w, scale, shift, out, B, H, W);
    else if (C == 12 && Cout == 32) hipLaunchKernelGGL((k_conv_first<12, 32>), grid, block, 0, s, obs, w, scale, shift, out, B, H, W);
}

void lz_launch_avgpool(const float *in, float *out, int B, int Hin, int Win, int C, hipStream_t s)
{
    const int Ho = (Hin + 1) / 2, Wo = (Win + 1) / 2;
    const int64_t n = (int64_t)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(k_avgpool, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, B, Hin, Win, C);
}

void lz_launch_conv1x1(const float *in, const float *w, const float *bias, const float *scale, const float *shift,
                       float *out, int B, int HW, int CIN, int Cout, hipStream_t s)
{
    const int npix = B * HW;
    if (CIN == 64) hipLaunchKernelGGL((k_conv1x1<64>), dim3((npix + 63) / 64), dim3(256), 0, s, in, w, bias, scale, shift, out, npix, Cout);
}

void lz_launch_lstm(const lz_lstm_args &a, hipStream_t s)
{
    const size_t lds = (size_t)(2 * 64 * 68 + 2 * 32 * 68) * 4;  // 52,224 B >= red (32 KB) + csum (8.4 KB)
    dim3 grid((a.B + 63) / 64, (4 * a.H) / 32), block(256);
    hipLaunchKernelGGL(k_lstm, grid, block, lds, s, a);
}

void lz_launch_heads(const lz_head_desc *heads, int nheads, int B, int HW, int C, int HC, int HID, hipStream_t s)
{
    head_pack hp;
    int k1max = 0;
    bool conv = false;
    for (int i = 0; i < nheads && i < MAXH; ++i) {
        hp.h[i] = heads[i];
        if (heads[i].K1 > k1max) k1max = heads[i].K1;
        conv |= heads[i].has_conv != 0;
    }
    size_t lds = ((size_t)EPB * k1max + (size_t)EPB * HID + 8 + (conv ? (size_t)EPB * HW * (C + 1) : 0)) * 4;
    hipLaunchKernelGGL(k_heads, dim3((B + EPB - 1) / EPB, nheads), dim3(256), lds, s, hp, B, HW, C, HC, HID);
}
