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
#include <stdio.h>
#include <stdlib.h>

#include "lz_nn_kernels.h"
#include "lz_hinv.h"
#define LZ_TREE_DEV_RESTORE_FAST_CONTRACT
#include "lz_tree_dev.h"
#include "lz_nn_dev.h"

namespace {

// (the reference's class mixes them: its representation network keeps ReLU, its dynamics network takes the model's activation, its prediction
// network keeps GELU -- sampled_efficientzero_model.py:177-218 passes `activation` to the dynamics network only -- so the chain's GELU instance
// reads a code per layer and per 1x1 job)

// ------------------------------------------------------------------------------------------------
// 3x3 convolution as implicit GEMM.  grid = (ceil(B*Hout*Wout / 144), Cout / 16), block = 256.
// ------------------------------------------------------------------------------------------------
template <int CIN, int STRIDE>
__global__ __launch_bounds__(256) void k_conv3x3(lz_conv_args a, int npix_max)
{
    constexpr int VEC = CIN / 16;   // floats per lane per operand fetch (64 -> b128, 32 -> b64)
    constexpr int PS = CIN + 4;     // padded pixel stride (floats): conflict-free fragment reads
    constexpr int TM = 144, MT = 9;
    constexpr int CH4 = CIN / 4;
    typedef typename vecf<VEC>::type vec_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sA = smem;                                   // [npix_max + 1][PS]; the extra pixel is all zeros
    float *sB = smem + (size_t)(npix_max + 1) * PS;     // [9][16][PS]
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

    // ---- stage the input halo (a contiguous pixel range of the NHWC tensor, optionally gathered per root from
    // a pool) and the 16-channel weight slice.  Loads are issued in batches of 8 per thread so that the whole
    // tile is in flight after 2-3 round trips instead of one round trip per 16 bytes.
    {
        const int nA = npix * CH4, nB = 9 * 16 * CH4, nTot = nA + nB;
        const float *wsrc = a.w + (size_t)blockIdx.y * 9 * 16 * CIN;
        constexpr int UB = 8;
        for (int base0 = 0; base0 < nTot; base0 += UB * 256) {
            float4 v[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int idx = base0 + u * 256 + tid;
                v[u] = vzero4();
                if (idx < nA) {
                    const int pix = idx / CH4, c4 = idx - pix * CH4;
                    const int q = in_lo + pix;
                    const float *src = a.in + (size_t)q * CIN + c4 * 4;
                    if (a.gather_ix) src += (size_t)a.gather_ix[q / HWin] * a.slot_stride;
                    v[u] = *reinterpret_cast<const float4 *>(src);
                } else if (idx < nTot) {
                    const int j = idx - nA;
                    v[u] = *reinterpret_cast<const float4 *>(wsrc + (size_t)j * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int idx = base0 + u * 256 + tid;
                if (idx < nA) {
                    const int pix = idx / CH4, c4 = idx - pix * CH4;
                    *reinterpret_cast<float4 *>(sA + (size_t)pix * PS + c4 * 4) = v[u];
                } else if (idx < nTot) {
                    const int j = idx - nA, row = j / CH4, c4 = j - row * CH4;
                    *reinterpret_cast<float4 *>(sB + (size_t)row * PS + c4 * 4) = v[u];
                }
            }
        }
        if (tid < CH4) *reinterpret_cast<float4 *>(sA + (size_t)npix_max * PS + tid * 4) = vzero4();
    }
    // ---- per-lane geometry of its row in each of the 9 M-tiles: LDS offset of the centre tap and a 9-bit
    // mask of the taps that fall inside the image (masked taps read the all-zero pixel)
    const int zoff = npix_max * PS;
    int base[MT], mask[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + i * 16 + (lane & 15);
        const int mm = min(m, m1);
        const int b = mm / HWout, p = mm - b * HWout, y = p / a.Wout, x = p - y * a.Wout;
        const int cy = STRIDE * y, cx = STRIDE * x;
        int mk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = cy + t / 3 - 1, ix = cx + t % 3 - 1;
            const int ok = (iy >= 0) & (iy < a.Hin) & (ix >= 0) & (ix < a.Win) & (m <= m1);
            mk |= ok << t;
        }
        int bs = ((b * a.Hin + cy) * a.Win + cx - in_lo) * PS;
        // keep the tap mask an opaque integer: otherwise the compiler turns it into 81 separate lane-mask
        // predicates that spill out of the SGPR file
        asm volatile("" : "+v"(mk), "+v"(bs));
        base[i] = bs;
        mask[i] = mk;
    }
    __syncthreads();

    // ---- K loop: wave wv owns channel group wv (16 or 8 channels) of every tap.  Operand fragments of tap t+1
    // are fetched while the MFMAs of tap t issue; consecutive MFMAs go to different accumulators.
    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int coff = wv * 4 * VEC + (lane >> 4) * VEC;
    const float *sBl = sB + (size_t)(lane & 15) * PS + coff;
    const float *sAl = sA + coff;
    vec_t af[2][MT], bf[2];
    auto fetch = [&](int t, int buf) {
        const int toff = ((t / 3 - 1) * a.Win + (t % 3 - 1)) * PS;
        bf[buf] = *reinterpret_cast<const vec_t *>(sBl + (size_t)t * 16 * PS);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int bit = (mask[i] >> t) & 1;
            const int off = zoff + bit * (base[i] + toff - zoff);
            af[buf][i] = *reinterpret_cast<const vec_t *>(sAl + off);
        }
    };
    fetch(0, 0);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int cur = t & 1;
        if (t + 1 < 9) fetch(t + 1, cur ^ 1);
#pragma unroll
        for (int j = 0; j < VEC; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(vget(af[cur][i], j), vget(bf[cur], j), acc[i], 0, 0, 0);
    }
    // ---- cross-wave (split-K) reduction through LDS, fused epilogue
    const int l = tid & 63, r = tid >> 6;
    const int col = l & 15, row_in_tile = 4 * (l >> 4) + r;  // C/D layout of mfma_f32_16x16x4
    const int co = n0 + col;
    const float sc = a.scale[co], sh = a.shift[co];
    float extra[MT];  // action-table and residual terms, fetched before the barrier so their latency overlaps it
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = min(m0 + i * 16 + row_in_tile, m1);
        const int b = m / HWout, p = m - b * HWout;
        float tab = 0.0f, res = 0.0f;
        if (a.act_table) tab = a.act_table[((size_t)a.action[b] * HWout + p) * a.Cout + co];
        if (a.residual) {
            const float *rp = a.residual;
            if (a.residual_gather) rp += (size_t)a.gather_ix[b] * a.slot_stride;
            res = rp[(size_t)m * a.Cout + co];
        }
        extra[i] = tab * sc + res;  // (v + tab)*sc + sh + res  ==  v*sc + (tab*sc + res) + sh  up to rounding
    }
    __syncthreads();
    float *red = smem;  // [4][MT][4][64]
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[((wv * MT + i) * 4 + q) * 64 + lane] = acc[i][q];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + i * 16 + row_in_tile;
        if (m > m1) continue;
        float v = red[((0 * MT + i) * 4 + r) * 64 + l] + red[((1 * MT + i) * 4 + r) * 64 + l] +
                  red[((2 * MT + i) * 4 + r) * 64 + l] + red[((3 * MT + i) * 4 + r) * 64 + l];
        v = v * sc + sh + extra[i];
        if (a.relu) v = fmaxf(v, 0.0f);
        a.out[(size_t)m * a.Cout + co] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// 3x3 convolution for the large grids of the representation tower (48x48 .. 12x12): one workgroup = TM output
// pixels x ALL output channels, wave = (16-channel N-tile, M-group), weight fragments streamed from L2 through a
// register ring (no split-K, no LDS round trip for the result), halo staged once in LDS.  Several workgroups are
// resident per CU so one's staging overlaps another's MFMAs.
// grid = ceil(B*Hout*Wout / TM), block = 256.
// ------------------------------------------------------------------------------------------------
template <int CIN, int COUT, int STRIDE, int TM>
__global__ __launch_bounds__(256) void k_conv3x3_big(lz_conv_args a, int npix_max)
{
    constexpr int PS = CIN + 4, CH4 = CIN / 4, G = CIN / 16, NT = COUT / 16, MG = 4 / NT;
    constexpr int MTW = TM / 16 / MG;      // M-tiles per wave
    constexpr int STEPS = 9 * G, R = (STEPS >= 12) ? 12 : 6;
    static_assert(TM % (16 * MG) == 0, "tile split");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sA = smem;  // [npix_max + 1][PS], last pixel all zeros
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nt = wv % NT, mg = wv / NT;
    const int HWin = a.Hin * a.Win, HWout = a.Hout * a.Wout;
    const int M = a.B * HWout;
    const int m0 = blockIdx.x * TM;
    const int m1 = min(m0 + TM, M) - 1;
    auto centre = [&](int m) -> int {
        if (STRIDE == 1) return m;
        const int b = m / HWout, p = m - b * HWout, y = p / a.Wout, x = p - y * a.Wout;
        return (b * a.Hin + STRIDE * y) * a.Win + STRIDE * x;
    };
    const int in_lo = max(0, centre(m0) - a.Win - 1);
    const int in_hi = min(a.B * HWin, centre(m1) + a.Win + 2);
    const int npix = in_hi - in_lo;
    // weight ring prologue first: its latency overlaps the halo staging
    const f32x4 *wl = reinterpret_cast<const f32x4 *>(a.wf) + (size_t)nt * STEPS * 64 + lane;
    f32x4 wq[R];
#pragma unroll
    for (int s = 0; s < R; ++s) wq[s] = wl[s * 64];
    {
        const int nA = npix * CH4;
        constexpr int UB = 8;
        for (int base0 = 0; base0 < nA; base0 += UB * 256) {
            f32x4 v[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int idx = min(base0 + u * 256 + tid, nA - 1);
                v[u] = *reinterpret_cast<const f32x4 *>(a.in + ((size_t)in_lo * CIN) + (size_t)idx * 4);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int idx = base0 + u * 256 + tid;
                if (idx < nA) *reinterpret_cast<f32x4 *>(sA + (size_t)(idx / CH4) * PS + (idx % CH4) * 4) = v[u];
            }
        }
        if (tid < CH4) *reinterpret_cast<f32x4 *>(sA + (size_t)npix_max * PS + tid * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int zoff = npix_max * PS;
    int base[MTW], mask[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int m = m0 + (mg * MTW + i) * 16 + (lane & 15);
        const int mm = min(m, m1);
        const int b = mm / HWout, p = mm - b * HWout, y = p / a.Wout, x = p - y * a.Wout;
        const int cy = STRIDE * y, cx = STRIDE * x;
        int mk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = cy + t / 3 - 1, ix = cx + t % 3 - 1;
            mk |= ((iy >= 0) & (iy < a.Hin) & (ix >= 0) & (ix < a.Win) & (m <= m1)) << t;
        }
        int bs = ((b * a.Hin + cy) * a.Win + cx - in_lo) * PS;
        asm volatile("" : "+v"(mk), "+v"(bs));
        base[i] = bs;
        mask[i] = mk;
    }
    __syncthreads();
    const float *sAl = sA + (lane >> 4) * 4;
    f32x4 acc[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto fetch_a = [&](int s, f32x4 (&af)[MTW]) {
        const int t = s / G, g = s % G;
        const int toff = ((t / 3 - 1) * a.Win + (t % 3 - 1)) * PS;
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            const int bit = (mask[i] >> t) & 1;
            const int off = zoff + bit * (base[i] + toff - zoff);
            af[i] = *reinterpret_cast<const f32x4 *>(sAl + off + g * 16);
        }
    };
    // the epilogue's operands do not depend on the accumulators: request them before the K loop (unconditional, clamped
    // loads; a load inside the epilogue's per-element conditions costs one exposed memory round trip each)
    const int col = nt * 16 + (lane & 15);
    const float sc = a.scale[col], sh = a.shift[col];
    const float *resp = a.residual ? a.residual : a.out;  // dummy: valid memory, switched off by a select
    const bool has_res = a.residual != nullptr, relu = a.relu != 0;
    float rv[MTW][4];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = min(m0 + (mg * MTW + i) * 16 + 4 * (lane >> 4) + q, m1);
            rv[i][q] = resp[(size_t)m * COUT + col];
        }
    f32x4 af[2][MTW];
    fetch_a(0, af[0]);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const f32x4 bfr = wq[s % R];
        if (s + R < STEPS) wq[s % R] = wl[(s + R) * 64];
        if (s + 1 < STEPS) fetch_a(s + 1, af[(s + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < MTW; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s & 1][i][j], bfr[j], acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = m0 + (mg * MTW + i) * 16 + 4 * (lane >> 4) + q;
            float v = acc[i][q] * sc + sh;
            v += has_res ? rv[i][q] : 0.0f;
            v = relu ? fmaxf(v, 0.0f) : v;
            if (m <= m1) a.out[(size_t)m * COUT + col] = v;
        }
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride-1 convolution of the representation tower by Winograd F(2x2, 3x3) (Lavin & Gray 2016): an output tile of 2x2
// pixels comes from a 4x4 input patch through 16 independent [tiles x CIN] x [CIN x COUT] products (one per transform point)
// instead of 36 -- 2.25x fewer MFMAs, paid with an input transform V = B^T d B (adds only), pre-transformed weights
// U = G g G^T (host, lz_model.h) and an output transform Y = A^T M A that happens entirely inside a lane: the 16x16x4 MFMA's
// C layout keeps a (tile, channel) element of every point in the same lane and slot.
// One workgroup = TMT consecutive output tiles (4 TMT pixels) x all COUT; wave = (16-channel N-tile, M-group); the transformed
// patches go through LDS in two chunks of 8 points (35 KB for CIN = 64, TMT = 16); weight fragments stream from L2
// through the same 12-deep register ring as k_conv3x3_big.  fp32 throughout; rounding differs from the direct form at the 1e-6
// level (tests/test_nn_golden_gpu.py re-qualifies the tower against the reference modules' outputs at 2e-5).
// grid = ceil(B * (H/2) * (W/2) / 32), block = 256.  H and W even.
// ------------------------------------------------------------------------------------------------
template <int CIN, int COUT, int TMT>
__global__ __launch_bounds__(256) void k_conv_wino(lz_conv_args a)
{
    constexpr int PS = CIN + 4, G = CIN / 16, NT = COUT / 16, MG = 4 / NT, MTW = (TMT / 16) / MG;
    constexpr int PCH = 8, Q4 = CIN / 4, IPT = TMT * Q4 / 256, STEPS = 16 * G, R = 12;
    static_assert(IPT >= 1 && MTW >= 1 && TMT * Q4 % 256 == 0, "tile split");
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [PCH][TMT][PS]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nt = wv % NT, mg = wv / NT;
    const int H = a.Hout, W = a.Wout, TH = H / 2, TW = W / 2, tpi = TH * TW, ntiles = a.B * tpi;
    const int t0 = blockIdx.x * TMT;
    const f32x4 *wl = reinterpret_cast<const f32x4 *>(a.uf) + (size_t)nt * STEPS * 64 + lane;
    f32x4 wq[R];
#pragma unroll
    for (int s = 0; s < R; ++s) wq[s] = wl[s * 64];
    // ---- input transform: item = (tile, channel quad); every load unconditional (clamped), borders zeroed by selects
    f32x4 V[IPT][16];
#pragma unroll
    for (int it = 0; it < IPT; ++it) {
        const int item = tid + 256 * it, tl = item / Q4, c4 = item % Q4;
        const int t = min(t0 + tl, ntiles - 1);
        const int b = t / tpi, r = t - b * tpi, ty = r / TW, tx = r - ty * TW;
        f32x4 d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int y = 2 * ty - 1 + i, x = 2 * tx - 1 + j;
                const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
                d[i][j] = *reinterpret_cast<const f32x4 *>(a.in + ((size_t)(b * H + yc) * W + xc) * CIN + c4 * 4);
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int y = 2 * ty - 1 + i, x = 2 * tx - 1 + j;
                const bool ok = y >= 0 && y < H && x >= 0 && x < W;
                d[i][j] = ok ? d[i][j] : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        f32x4 e[4][4];  // B^T d
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            e[0][j] = d[0][j] - d[2][j];
            e[1][j] = d[1][j] + d[2][j];
            e[2][j] = d[2][j] - d[1][j];
            e[3][j] = d[1][j] - d[3][j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // (B^T d) B
            V[it][4 * i + 0] = e[i][0] - e[i][2];
            V[it][4 * i + 1] = e[i][1] + e[i][2];
            V[it][4 * i + 2] = e[i][2] - e[i][1];
            V[it][4 * i + 3] = e[i][1] - e[i][3];
        }
    }
    const int col = nt * 16 + (lane & 15);
    const float sc = a.scale[col], sh = a.shift[col];
    const bool has_res = a.residual != nullptr, relu = a.relu != 0;
    f32x4 acc[16][MTW];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int m = 0; m < MTW; ++m) acc[p][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float *sA = smem + (size_t)((mg * MTW) * 16 + (lane & 15)) * PS + (lane >> 4) * 4;
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        if (ch) __syncthreads();  // every wave has read the previous chunk
#pragma unroll
        for (int it = 0; it < IPT; ++it) {
            const int item = tid + 256 * it, tl = item / Q4, c4 = item % Q4;
#pragma unroll
            for (int pp = 0; pp < PCH; ++pp)
                *reinterpret_cast<f32x4 *>(smem + (size_t)(pp * TMT + tl) * PS + c4 * 4) = V[it][ch * PCH + pp];
        }
        __syncthreads();
#pragma unroll
        for (int pp = 0; pp < PCH; ++pp) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                        const int s = (ch * PCH + pp) * G + g;
                const f32x4 bfr = wq[s % R];
                if (s + R < STEPS) wq[s % R] = wl[(s + R) * 64];
                f32x4 af[MTW];
#pragma unroll
                for (int m = 0; m < MTW; ++m) af[m] = *reinterpret_cast<const f32x4 *>(sA + (size_t)(pp * TMT + m * 16) * PS + g * 16);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
                        acc[ch * PCH + pp][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][j], bfr[j], acc[ch * PCH + pp][m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // ---- output transform (per lane), BN (+ residual) (+ ReLU), 2x2 pixels per tile
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int tl = (mg * MTW + m) * 16 + 4 * (lane >> 4) + q;
            const int t = t0 + tl;
            float M[16];
#pragma unroll
            for (int p = 0; p < 16; ++p) M[p] = acc[p][m][q];
            float sA_[2][4];  // A^T M
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sA_[0][j] = (M[0 * 4 + j] + M[1 * 4 + j]) + M[2 * 4 + j];
                sA_[1][j] = (M[1 * 4 + j] - M[2 * 4 + j]) - M[3 * 4 + j];
            }
            if (t < ntiles) {
                const int b = t / tpi, r = t - b * tpi, ty = r / TW, tx = r - ty * TW;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    const float y0 = (sA_[dy][0] + sA_[dy][1]) + sA_[dy][2];
                    const float y1 = (sA_[dy][1] - sA_[dy][2]) - sA_[dy][3];
                    const size_t o = ((size_t)(b * H + 2 * ty + dy) * W + 2 * tx) * COUT + col;
                    float v0 = y0 * sc + sh, v1 = y1 * sc + sh;
                    if (has_res) { v0 += a.residual[o]; v1 += a.residual[o + COUT]; }
                    if (relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); }
                    a.out[o] = v0;
                    a.out[o + COUT] = v1;
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------
// first DownSample layer: conv3x3 / stride 2 from NCHW observations, + BN + ReLU.  One thread per
// output pixel computes all Cout channels from the (<= 9*C)-value patch; weights broadcast from LDS.
// ------------------------------------------------------------------------------------------------

// OUTBF (fast mode): the output tensor is bf16 NHWC (the tower keeps its activations in bf16 there)
template <int C, int COUT, bool OUTBF = false>
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
    // all 9 x C input values are requested before the first use, unconditionally (clamped coordinates, zeroed by a select
    // afterwards): a predicated load per tap is a branch plus a full wait each, i.e. nine dependent HBM round trips
    float xv[9][C];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int iy = 2 * y + t / 3 - 1, ix = 2 * x + t % 3 - 1;
        const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
#pragma unroll
        for (int ci = 0; ci < C; ++ci) xv[t][ci] = obs[(((size_t)b * C + ci) * H + cy) * W + cx];
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int iy = 2 * y + t / 3 - 1, ix = 2 * x + t % 3 - 1;
        const bool v = iy >= 0 && iy < H && ix >= 0 && ix < W;
#pragma unroll
        for (int ci = 0; ci < C; ++ci) {
            const float xs = v ? xv[t][ci] : 0.0f;
            const float *wr = sw + (t * C + ci) * COUT + g * 8;
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] += xs * wr[c];
        }
    }
    const float *sc = scale + g * 8, *sh = shift + g * 8;
    if constexpr (OUTBF) {
        bf16x8 v;
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = (__bf16)fmaxf(acc[c] * sc[c] + sh[c], 0.f);
        *reinterpret_cast<bf16x8 *>(reinterpret_cast<__bf16 *>(out) + (size_t)m * COUT + g * 8) = v;
        return;
    }
    float *o = out + (size_t)m * COUT + g * 8;
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

// The same layer for 4-channel observations on the matrix pipe (round 4): a workgroup owns TR = 96 / Wo output rows of one image = 96 output
// pixels (six 16-pixel MFMA column tiles); the 2 TR + 1 input rows of the four NCHW planes it needs arrive as whole coalesced rows (16-byte
// loads) in LDS [plane][row][4 + W] -- the VALU kernel above reads its 36 inputs per thread as stride-2 four-byte loads, four threads per pixel
// re-reading them.  k = the 4 input planes of one tap (v_mfma_f32_16x16x4_f32, weights as the A operand: D[channel][pixel]), 9 taps accumulate;
// a lane ends with four consecutive output channels of one pixel.  Wave (nt, mg): output-channel tile nt, pixel tiles 3 mg .. 3 mg + 2.
template <bool OUTBF>
__global__ __launch_bounds__(256) void k_conv_first_mm(const float *__restrict__ obs, const float *__restrict__ w, const float *__restrict__ scale,
                                                       const float *__restrict__ shift, float *__restrict__ out, int B, int H, int W, int TR)
{
    constexpr int C = 4, COUT = 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [C][2 TR + 1][W + 8]: column 4 + ix (column 3 = ix -1, zero)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nt = wv & 1, mg = wv >> 1;
    const int Ho = H / 2, Wo = W / 2, bands = (Ho + TR - 1) / TR, NR = 2 * TR + 1, WP = W + 8;
    const int img = blockIdx.x / bands, band = blockIdx.x - img * bands, oy0 = band * TR, iy0 = 2 * oy0 - 1;
    // ---- weights: lane (co = 16 nt + (l & 15), ci = l >> 4) of tap t: w[t][ci][co]
    float wq[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wq[t] = w[(t * C + (lane >> 4)) * COUT + nt * 16 + (lane & 15)];
    const int co4 = nt * 16 + 4 * (lane >> 4);
    const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + co4), sh = *reinterpret_cast<const f32x4 *>(shift + co4);
    // ---- stage the rows (zero outside the image)
    {
        const int W4 = W / 4, n4 = C * NR * W4;
        const float *src = obs + (size_t)img * C * H * W;
        for (int i0 = 0; i0 < n4; i0 += 2 * 256) {
            f32x4 v[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = min(i0 + u * 256 + tid, n4 - 1), c4 = idx % W4, rr = idx / W4, r = rr % NR, ci = rr / NR, iy = iy0 + r;
                const f32x4 t = *reinterpret_cast<const f32x4 *>(src + ((size_t)ci * H + min(max(iy, 0), H - 1)) * W + c4 * 4);
                v[u] = (iy >= 0 && iy < H) ? t : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = i0 + u * 256 + tid;
                if (idx < n4) {
                    const int c4 = idx % W4, rr = idx / W4;
                    *reinterpret_cast<f32x4 *>(smem + rr * WP + 4 + c4 * 4) = v[u];
                }
            }
        }
        if (tid < C * NR) smem[tid * WP + 3] = 0.0f;   // ix = -1
    }
    __syncthreads();
    // ---- products
    f32x4 acc[3];
    int pbase[3], prow[3], pcol[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int p = 16 * (3 * mg + i) + (lane & 15);
        prow[i] = p / Wo; pcol[i] = p - prow[i] * Wo;
        pbase[i] = ((lane >> 4) * NR + 2 * prow[i]) * WP + 3 + 2 * pcol[i];   // plane l >> 4, tap (0, 0)
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float px[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) px[i] = smem[pbase[i] + (t / 3) * WP + (t % 3)];
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[t], px[i], acc[i], 0, 0, 0);
    }
    // ---- BatchNorm + ReLU, four consecutive channels of one pixel per lane
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (oy0 + prow[i] >= Ho) continue;
        const size_t o = (((size_t)img * Ho + oy0 + prow[i]) * Wo + pcol[i]) * COUT + co4;
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fmaxf(acc[i][q] * sc[q] + sh[q], 0.0f);
        if constexpr (OUTBF) {
            bf16x4 hb;
#pragma unroll
            for (int q = 0; q < 4; ++q) hb[q] = (__bf16)v[q];
            *reinterpret_cast<bf16x4 *>(reinterpret_cast<__bf16 *>(out) + o) = hb;
        } else {
            *reinterpret_cast<f32x4 *>(out + o) = v;
        }
    }
}

// first layer of a no-downsample RepresentationNetwork (board games, common.py:735-741,768-771): conv3x3 stride 1 from
// NCHW observations [B][C][H][W] to NHWC [B][H*W][64], + BN + ReLU.  thread = (pixel, 8 output channels).
// COUT = 64 | 32 | 16 output channels: COUT / 8 threads per pixel
template <int COUT>
__global__ __launch_bounds__(256) void k_conv_in(const float *__restrict__ obs, const float *__restrict__ w,
                                                 const float *__restrict__ scale, const float *__restrict__ shift,
                                                 float *__restrict__ out, int B, int C, int H, int W)
{
    constexpr int TPP = COUT / 8, PPB = 256 / TPP;  // threads per pixel, pixels per block
    extern __shared__ float sw[];  // [9][C][COUT]
    for (int i = threadIdx.x; i < 9 * C * COUT; i += 256) sw[i] = w[i];
    __syncthreads();
    const int g = threadIdx.x % TPP;
    const int64_t m = (int64_t)blockIdx.x * PPB + (threadIdx.x / TPP);
    if (m >= (int64_t)B * H * W) return;
    const int b = (int)(m / (H * W)), p = (int)(m - (int64_t)b * H * W), y = p / W, x = p - y * W;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.0f;
    for (int t = 0; t < 9; ++t) {
        const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        for (int ci = 0; ci < C; ++ci) {
            const float xv = obs[(((size_t)b * C + ci) * H + iy) * W + ix];
            const float *wr = sw + (t * C + ci) * COUT + g * 8;
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] += xv * wr[c];
        }
    }
    float *o = out + (size_t)m * COUT + g * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = fmaxf(acc[c] * scale[g * 8 + c] + shift[g * 8 + c], 0.0f);
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

// the same pooling on a bf16 NHWC input (fast mode), 8 channels per thread; OUTBF: bf16 output, otherwise fp32 (the chain's input)
template <bool OUTBF>
__global__ __launch_bounds__(256) void k_avgpool_bf(const __bf16 *__restrict__ in, void *__restrict__ out, int B, int Hin, int Win, int C)
{
    const int Ho = (Hin + 1) / 2, Wo = (Win + 1) / 2, C8 = C / 8;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)B * Ho * Wo * C8) return;
    const int c8 = (int)(idx % C8);
    const int64_t pix = idx / C8;
    const int x = (int)(pix % Wo), y = (int)((pix / Wo) % Ho), b = (int)(pix / ((int64_t)Wo * Ho));
    float sacc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) sacc[c] = 0.0f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int iy = 2 * y + t / 3 - 1, ix = 2 * x + t % 3 - 1;
        if (iy >= 0 && iy < Hin && ix >= 0 && ix < Win) {
            const bf16x8 v = *reinterpret_cast<const bf16x8 *>(in + (((size_t)b * Hin + iy) * Win + ix) * C + c8 * 8);
#pragma unroll
            for (int c = 0; c < 8; ++c) sacc[c] += (float)v[c];
        }
    }
    if constexpr (OUTBF) {
        bf16x8 o;
#pragma unroll
        for (int c = 0; c < 8; ++c) o[c] = (__bf16)(sacc[c] / 9.0f);
        *reinterpret_cast<bf16x8 *>(reinterpret_cast<__bf16 *>(out) + (size_t)pix * C + c8 * 8) = o;
    } else {
        float *o = reinterpret_cast<float *>(out) + (size_t)pix * C + c8 * 8;
        *reinterpret_cast<float4 *>(o) = make_float4(sacc[0] / 9.0f, sacc[1] / 9.0f, sacc[2] / 9.0f, sacc[3] / 9.0f);
        *reinterpret_cast<float4 *>(o + 4) = make_float4(sacc[4] / 9.0f, sacc[5] / 9.0f, sacc[6] / 9.0f, sacc[7] / 9.0f);
    }
}

// fast mode, 64x64 observations: the tower ends without a second pooling (common.py:358-359), so its bf16 output becomes the chain's fp32 input here
__global__ __launch_bounds__(256) void k_bf16_to_f32(const __bf16 *__restrict__ in, float *__restrict__ out, size_t n8)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const bf16x8 v = *reinterpret_cast<const bf16x8 *>(in + i * 8);
    *reinterpret_cast<float4 *>(out + i * 8) = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    *reinterpret_cast<float4 *>(out + i * 8 + 4) = make_float4((float)v[4], (float)v[5], (float)v[6], (float)v[7]);
}

// conv1x1 (64 -> 16) + bias + BN + ReLU as a small MFMA GEMM: 144 pixels x 16 channels per workgroup, the 4
// waves split K = 64 into 16-channel groups (one operand fetch + 4 MFMAs per tile per wave).
__global__ __launch_bounds__(256) void k_conv1x1(lz_c1_args a)
{
    constexpr int CIN = 64, PS = CIN + 4, MT = 9, TM = 144;
    __shared__ __attribute__((aligned(16))) float smem[(TM + 16) * PS];  // A [144][68] + W [16][68]; reused for the reduction
    const lz_c1_job &jb = a.job[blockIdx.y];
    float *sA = smem, *sB = smem + TM * PS;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int m0 = blockIdx.x * TM, m1 = min(m0 + TM, a.npix) - 1;
    {
        float4 v[10];
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            const int idx = u * 256 + tid;
            v[u] = vzero4();
            if (idx < TM * 16) {
                const int pix = idx >> 4, c4 = idx & 15;
                if (m0 + pix <= m1) v[u] = *reinterpret_cast<const float4 *>(jb.in + (size_t)(m0 + pix) * CIN + c4 * 4);
            } else if (idx < (TM + 16) * 16) {
                v[u] = *reinterpret_cast<const float4 *>(jb.w + (size_t)(idx - TM * 16) * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            const int idx = u * 256 + tid;
            if (idx < (TM + 16) * 16) *reinterpret_cast<float4 *>(smem + (size_t)(idx >> 4) * PS + (idx & 15) * 4) = v[u];
        }
    }
    __syncthreads();
    const int coff = wv * 16 + (lane >> 4) * 4;
    const float4 bf = *reinterpret_cast<const float4 *>(sB + (lane & 15) * PS + coff);
    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const float4 af = *reinterpret_cast<const float4 *>(sA + (i * 16 + (lane & 15)) * PS + coff);
        acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(vget(af, j), vget(bf, j), acc[i], 0, 0, 0);
    }
    __syncthreads();
    float *red = smem;  // [4][MT][4][64] = 9216 floats <= (144+16)*68
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[((wv * MT + i) * 4 + q) * 64 + lane] = acc[i][q];
    __syncthreads();
    const int l = tid & 63, r = tid >> 6, col = l & 15, row_in_tile = 4 * (l >> 4) + r;
    const float bi = jb.bias[col], sc = jb.scale[col], sh = jb.shift[col];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + i * 16 + row_in_tile;
        if (m > m1) continue;
        float v = red[((0 * MT + i) * 4 + r) * 64 + l] + red[((1 * MT + i) * 4 + r) * 64 + l] +
                  red[((2 * MT + i) * 4 + r) * 64 + l] + red[((3 * MT + i) * 4 + r) * 64 + l];
        v = (v + bi) * sc + sh;
        jb.out[(size_t)m * jb.out_stride + jb.out_off + col] = fmaxf(v, 0.0f);
    }
}

// ------------------------------------------------------------------------------------------------
// Convolution chain on the 6x6 latent: one workgroup per root, wave w owns output channels [16w, 16w+16) of every
// layer (3 M-tiles of 16 pixels, the last 12 rows are padding), K = 9 taps x 64 channels = 36 steps of 12 MFMAs.
// Activations ping-pong between four LDS buffers; weight fragments come straight from L2 into registers, four
// steps ahead.  432 MFMAs per wave per layer = 5.8 us at the fp32-matrix issue rate.
// ------------------------------------------------------------------------------------------------
// TREE != 0: wave 0 first runs the root's tree step (dev_step_lds, variant TREE - 1) on an LDS copy of the tree placed in
// the still unused activation buffers -- the per-simulation tree launch disappears, and the weight prefetch of the first
// layer is in flight meanwhile.

template <int GW, int GH, bool TS = false, int TREE = 0>
__global__ __launch_bounds__(256) void k_chain(lz_chain_args a, typename step_arg<TREE>::type step)
{
    constexpr int PS = 68, HW = GW * GH, MT = (HW + 15) / 16, BUF = (HW + 1) * PS;  // HW pixels + one all-zero pixel
    extern __shared__ __attribute__((aligned(16))) float smem[];  // 4 activation buffers of BUF floats, then the
    float *sTab = smem + 4 * BUF;          // [HW][PS] one-hot-action table slice of this root's action
    float *sSS = sTab + HW * PS;           // [LZ_CHAIN_MAX_LAYERS][2][64] folded-BN scale / shift: the epilogues read LDS only,
                                           // a global load there would make the compiler drain the weight ring (vmcnt(0))
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.x;
    // first weight fragments are requested before anything else so that their L2 round trip overlaps the staging
    constexpr int R = 12;
    float4 wq[R];
    {
        const float4 *w0 = reinterpret_cast<const float4 *>(a.layer[0].wf) + (size_t)wv * 36 * 64 + lane;
#pragma unroll
        for (int s = 0; s < R; ++s) wq[s] = w0[s * 64];
    }
    // 1x1 head-conv weights of this wave's job too (used after the last layer)
    float4 c1w[4];
    {
        const float *cw = a.c1[min(wv, max(a.nc1 - 1, 0))].w;
#pragma unroll
        for (int g = 0; g < 4; ++g) c1w[g] = *reinterpret_cast<const float4 *>(cw + (size_t)(lane & 15) * 64 + g * 16 + (lane >> 4) * 4);
    }
    int g_slot = 0, g_action = 0;
    if constexpr (TREE != 0) {
        int32_t *s_sel = reinterpret_cast<int32_t *>(sSS + LZ_CHAIN_MAX_LAYERS * 128 + (TS ? 64 : 0));
        if (wv == 0)
            dev_step_lds<1, TREE - 1>(step.t, b, step.new_node, step.discount, step.vps, step.values, step.logits, step.horizon,
                                      step.a, step.delta, step.vtp, reinterpret_cast<float4 *>(smem), s_sel);
        __syncthreads();
        g_slot = s_sel[0];
        g_action = s_sel[1];
    } else {
        if (a.gather_ix) g_slot = a.gather_ix[b];
        if (a.act_table) g_action = a.action[b];
    }
    {
        const float *src = a.in + (size_t)b * HW * 64 + (size_t)g_slot * a.slot_stride;
        constexpr int NU = (HW * 16 + 255) / 256;
        float4 v[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int idx = u * 256 + tid;
            v[u] = vzero4();
            if (idx < HW * 16) v[u] = *reinterpret_cast<const float4 *>(src + (size_t)idx * 4);
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int idx = u * 256 + tid;
            if (idx < HW * 16) *reinterpret_cast<float4 *>(smem + (idx >> 4) * PS + (idx & 15) * 4) = v[u];
        }
        if (tid < 64) *reinterpret_cast<float4 *>(smem + (tid >> 4) * BUF + HW * PS + (tid & 15) * 4) = vzero4();
        if (a.act_table) {
            const float *tsrc = a.act_table + (size_t)g_action * HW * 64;
            float4 tv[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int idx = min(u * 256 + tid, HW * 16 - 1);
                tv[u] = *reinterpret_cast<const float4 *>(tsrc + (size_t)idx * 4);
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int idx = u * 256 + tid;
                if (idx < HW * 16) *reinterpret_cast<float4 *>(sTab + (idx >> 4) * PS + (idx & 15) * 4) = tv[u];
            }
        }
        for (int i = tid; i < a.nlayers * 128; i += 256) {
            const int L = i >> 7, r = i & 127;
            sSS[i] = (r < 64) ? a.layer[L].scale[r] : a.layer[L].shift[r - 64];
        }
    }
    // geometry of this lane's row in each of the MT M-tiles (the same for every layer)
    // 36 pixels = 2 full 16-row tiles + 4 rows (81 = 5 + 1 row): the remainder rows run on v_mfma_f32_4x4x1_f32
    // (16 blocks of 4 rows x 4 channels; block = (k-quarter, channel quad) -- exactly the lane layout the 16x16x4 B
    // fragment already has) at a quarter of the cost of a padded third tile.
    constexpr int MF = HW / 16, REM = HW % 16;
    constexpr bool SMALL_REM = REM > 0 && REM <= 4;
    static_assert(MT == MF + (REM ? 1 : 0), "tile count");
    const int zoff = HW * PS;
    int base[MT], mask[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int row = (SMALL_REM && i == MF) ? i * 16 + (lane & 3) : i * 16 + (lane & 15);
        const int p = min(row, HW - 1), y = p / GW, x = p - y * GW;
        int mk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
            mk |= ((iy >= 0) & (iy < GH) & (ix >= 0) & (ix < GW) & (row < HW)) << t;
        }
        int bs = p * PS;
        asm volatile("" : "+v"(mk), "+v"(bs));
        base[i] = bs;
        mask[i] = mk;
    }
    const int kq4 = (lane >> 4) * 4;
    // TS instantiation only (debugging): s_memtime stamps of workgroup 0 / thread 0 into LDS, copied out at the end
    unsigned long long *sTS = reinterpret_cast<unsigned long long *>(sSS + LZ_CHAIN_MAX_LAYERS * 128);
    int nts = 0;
#define LZ_TS() do { if constexpr (TS) { if (b == 0 && tid == 0) sTS[nts++] = __builtin_readcyclecounter(); } } while (0)
    LZ_TS();
    __syncthreads();
    LZ_TS();

    // Weight fragments are requested R = 12 steps (~4.6k MFMA cycles) before use into a register ring, across layer
    // boundaries (the next layer's first fragments are in flight during this layer's epilogue and barrier); the
    // A fragments of step s+1 are read from LDS before the MFMAs of step s issue.  sched_barrier pins that order
    // (left alone, the scheduler sinks the prefetches next to their uses and exposes the L2 latency).
    auto fetch_a = [&](const float *sIn, int s, float4 (&af)[MT]) {
        const int t = s >> 2, g = s & 3;
        const int toff = ((t / 3 - 1) * GW + (t % 3 - 1)) * PS;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int bit = (mask[i] >> t) & 1;
            const int off = zoff + bit * (base[i] + toff - zoff);
            af[i] = *reinterpret_cast<const float4 *>(sIn + off + g * 16);
        }
    };
    for (int L = 0; L < a.nlayers; ++L) {
        const lz_chain_layer &ly = a.layer[L];
        const float *sIn = smem + ly.in * BUF + kq4;
        float *sOut = smem + ly.out * BUF;
        const bool more = L + 1 < a.nlayers;
        const float4 *wc = reinterpret_cast<const float4 *>(ly.wf) + (size_t)wv * 36 * 64 + lane;
        const float4 *wn = reinterpret_cast<const float4 *>(a.layer[more ? L + 1 : L].wf) + (size_t)wv * 36 * 64 + lane;
        f32x4 acc[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // the epilogue's operands do not depend on the accumulators: read them now, their LDS latency hides under the K loop
        const int col = wv * 16 + (lane & 15);
        const float sc = sSS[L * 128 + col], sh = sSS[L * 128 + 64 + col];
        const bool tab = ly.act != 0, hasres = ly.res >= 0, relu = ly.relu != 0;
        const float *sRes = smem + max(ly.res, 0) * BUF;
        float tv[MT][4], rv[MT][4];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (SMALL_REM && i == MF && q > 0) continue;
                const int row = (SMALL_REM && i == MF) ? i * 16 + (lane >> 4) : i * 16 + 4 * (lane >> 4) + q;
                const int rr = min(row, HW - 1);
                tv[i][q] = sTab[rr * PS + col];
                rv[i][q] = sRes[rr * PS + col];
            }
        float4 af[2][MT];
        fetch_a(sIn, 0, af[0]);
        LZ_TS();
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            const float4 bfr = wq[s % R];
            wq[s % R] = (s + R < 36) ? wc[(s + R) * 64] : wn[(s + R - 36) * 64];
            if (s + 1 < 36) fetch_a(sIn, s + 1, af[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    if (SMALL_REM && i == MF) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(vget(af[s & 1][i], j), vget(bfr, j), acc[i], 0, 0, 0);
                    else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(vget(af[s & 1][i], j), vget(bfr, j), acc[i], 0, 0, 0);
                }
                // keep the short 4x4x1 instructions spread between the 16x16x4 ones: grouped at the end of the step (where
                // the scheduler puts them) each waits on its predecessor (s_nop), interleaved none does
                if (SMALL_REM) __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        LZ_TS();
        if (SMALL_REM) {
            // the 4x4x1 blocks hold partial sums per k-quarter (lane >> 4): add the four quarters, then lane group q keeps row q
            f32x4 r = acc[MF];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                r[q] = xor32_sum(xor16_sum(r[q]));
            }
            const int g = lane >> 4;
            const float mine = g == 0 ? r[0] : g == 1 ? r[1] : g == 2 ? r[2] : r[3];
            acc[MF] = (f32x4){mine, 0.f, 0.f, 0.f};
        }
        // epilogue: BN (+ action table) (+ residual) (+ ReLU) -> LDS (and the latent pool); branch-free, operands preloaded
        float outv[MT][4];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (SMALL_REM && i == MF && q > 0) continue;
                const int row = (SMALL_REM && i == MF) ? i * 16 + (lane >> 4) : i * 16 + 4 * (lane >> 4) + q;
                float v = acc[i][q];
                v += tab ? tv[i][q] : 0.0f;
                v = v * sc + sh;
                v += hasres ? rv[i][q] : 0.0f;
                v = relu ? fmaxf(v, 0.0f) : v;
                outv[i][q] = v;
                if (row < HW) sOut[row * PS + col] = v;
            }
        if (ly.gout) {
            float *go = ly.gout + (size_t)b * HW * 64 + col;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (SMALL_REM && i == MF && q > 0) continue;
                    const int row = (SMALL_REM && i == MF) ? i * 16 + (lane >> 4) : i * 16 + 4 * (lane >> 4) + q;
                    if (row < HW) go[(size_t)row * 64] = outv[i][q];
                }
        }
        LZ_TS();
        __syncthreads();
        LZ_TS();
    }
    // 1x1 head convolutions (64 -> 16) + bias + BN + ReLU: wave j runs job j
    if (wv < a.nc1) {
        const lz_c1_job &jb = a.c1[wv];
        const float *sIn = smem + a.c1_in[wv] * BUF + kq4;
        f32x4 acc[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bfr = c1w[g];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int row = i * 16 + (lane & 15);
                const int off = (row < HW) ? row * PS : zoff;
                const float4 af = *reinterpret_cast<const float4 *>(sIn + off + g * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(vget(af, j), vget(bfr, j), acc[i], 0, 0, 0);
            }
        }
        const int col = lane & 15;
        const float bi = jb.bias[col], sc = jb.scale[col], sh = jb.shift[col];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = i * 16 + 4 * (lane >> 4) + q;
                if (row < HW) {
                    const float v = (acc[i][q] + bi) * sc + sh;
                    jb.out[((size_t)b * HW + row) * jb.out_stride + jb.out_off + col] = fmaxf(v, 0.0f);
                }
            }
    }
    LZ_TS();
    if constexpr (TS) {
        if (b == 0 && tid == 0 && a.tstamp) {
            a.tstamp[0] = (unsigned long long)nts;
            for (int i = 0; i < nts && i < 30; ++i) a.tstamp[1 + i] = sTS[i];
        }
    }
#undef LZ_TS
}

// ------------------------------------------------------------------------------------------------
// The same chain by Winograd F(2x2, 3x3) (Lavin & Gray 2016) for even grids (6x6: 9 output tiles of 2x2 pixels per root).
// A layer = 16 independent [tiles x 64] x [64 x 64] products, one per transform point.  With 9 rows the 16-row MFMA would be
// 44 % padding, so the products run on v_mfma_f32_4x4x1_f32: 16 blocks of (4 tiles) x (4 channels) = 4 tiles x 64 output channels
// per instruction, one k per instruction, 3 row blocks (12 rows, 9 used): 16 x 64 x 3 x 8 = 6,144 matrix cycles per SIMD and layer
// instead of 10,368 for the direct form.  Wave (i, h) owns transform points 4i .. 4i+3 (row i of the 4x4 point grid) for ALL output
// channels and the h-th part of the input channels (NW = 4 waves: all of them; NW = 8: halves), so no partial sums meet inside a
// wave; the output transform Y = A^T M A is split the same way: a wave reduces its row (M[i][.] A, inside a lane), the row
// results meet in LDS and every thread finishes its (pixel, channel) outputs (BN, action table, residual, ReLU) -- written to LDS as
// the next layer's input and, for the latent, coalesced to the pool.
// Input transform V = B^T d B: one (tile, channel quad[, row pair]) item per thread, float4 LDS reads and writes, packed adds.
// Weights: U = G g G^T (host, binary64, rounded once; lz_model.h wino_chain) stream from L2 through a register ring that runs
// across layer boundaries: 262 KB per layer and workgroup (1.78x the direct form's bytes for 0.59x its matrix cycles) -- the
// layer loop is bound by what one CU can pull from L2 (tools/ubench/l2_stream.py: 33 B/clk with 4 waves, 43 with 8).
// fp32 throughout; rounding differs from the direct form at the 1e-6 level (tests/test_nn_golden_gpu.py qualifies the chain against
// the reference modules' outputs at 2e-5).  LZ_CHAIN_DIRECT=1 selects k_chain.
// ------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
// a - b on the packed-fp32 path (the compiler packs additions but leaves subtractions scalar)
__device__ __forceinline__ f32x4 pk_sub(f32x4 a, f32x4 b)
{
    f32x2 lo, hi;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(__builtin_shufflevector(a, a, 0, 1)), "v"(__builtin_shufflevector(b, b, 0, 1)));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(__builtin_shufflevector(a, a, 2, 3)), "v"(__builtin_shufflevector(b, b, 2, 3)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}



template <int GW, int GH, int NW, bool TS = false, int TREE = 0, int RING = (NW == 8 ? (GW * GH > 36 ? 8 : 16) : 32), bool HEADS = false, bool GELU = false>
__global__ __launch_bounds__(NW * 64) void k_chain_w(lz_chain_args a, typename step_arg<TREE>::type step)
{
    constexpr int PS = 68, HW = GW * GH, MT = (HW + 15) / 16, BUF = (HW + 1) * PS, NTHR = NW * 64;
    constexpr int TW = GW / 2, TH = GH / 2, NT = TW * TH, NRB = (NT + 3) / 4;
    constexpr int KH = NW / 4, KSW = 16 / KH, NSTEP = 4 * KSW, R = (RING < NSTEP) ? RING : NSTEP;  // k parts; channel quads per wave; steps; ring
    constexpr int NITEM = NT * 16 * KH;      // input-transform items: (tile, channel quad) x (row pair if 8 waves)
    constexpr int NOUT = (HW * 64 + NTHR - 1) / NTHR;  // outputs per thread in the combine step
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    static_assert((GW % 2) == 0 && (GH % 2) == 0 && NITEM <= NTHR && NT * 4 == HW && NRB <= 4, "even grid, one transform item per thread");
    extern __shared__ __attribute__((aligned(16))) float smem[];  // 4 activation buffers of BUF floats, then
    // BIG (8x8: 16 tiles): 160 KB hold four activation buffers and V only if the row results alias V (one more barrier per layer)
    // and the action-table slice is read from L2 by the one layer that needs it
    constexpr bool BIG = HW > 36;
    float *sTab = smem + 4 * BUF;                       // [HW][PS] one-hot-action table slice of this root's action (not BIG)
    float *sSS = sTab + (BIG ? 0 : HW * PS);            // [LZ_CHAIN_MAX_LAYERS][2][64] folded-BN scale / shift
    float *sMisc = sSS + LZ_CHAIN_MAX_LAYERS * 128;     // 128 floats: time stamps (TS) | the tree step's selection
    float *sV = sMisc + 128;                            // [16 points][NT][PS] transformed input patches
    float *sX = BIG ? sV : sV + 16 * NT * PS;           // [KH][4 point rows][NT][2][64] row results of the output transform
    static_assert(!BIG || NW * NT * 128 <= 16 * NT * PS, "the row results fit into V");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pi = wv & 3, kh = wv >> 2;
    const int b = blockIdx.x;
    lz_stamp_begin(a.stamp);
    // step s of a wave: point 4 pi + s / KSW, channel quad kh KSW + s % KSW; weights [16 points][16 quads][64 lanes] float4
    auto wofs = [](int s) { return ((s / KSW) * 16 + (s % KSW)) * 64; };
    const size_t wbase = (size_t)((4 * pi) * 16 + kh * KSW) * 64 + lane;
    f32x4 wq[R];
    auto fill_ring = [&]() {
        const f32x4 *w0 = reinterpret_cast<const f32x4 *>(a.layer[0].uc) + wbase;
#pragma unroll
        for (int s = 0; s < R; ++s) wq[s] = w0[wofs(s)];
    };
    // the first layer's weights are requested before anything else -- except with the tree step in the prologue: a whole layer of
    // fragments (256 registers) cannot stay live across it, so they are requested right after it, under the latent's staging
    if constexpr (TREE == 0) fill_ring();
    // 1x1 head convolutions at the end of the kernel: unit = (job, 16-pixel tile).  8 waves: wave j < nc1 runs tile 0 and the remainder
    // pixels (4 of 36, on the 4x4x1 instruction) of job j, wave nc1 + j its tile 1; 4 waves: wave j runs all tiles of job j.  Each wave
    // requests its job's weights and epilogue operands now (after the products they would be an exposed L2 round trip in the tail).
    constexpr bool C1SPLIT = NW == 8 && (HW % 16) != 0 && (HW % 16) <= 4 && HW / 16 == 2;
    const int nj = max(a.nc1, 1);
    const int c1j = C1SPLIT ? wv % nj : min(wv, nj - 1);   // this wave's job
    float4 c1w[4];
    float4 c1b, c1s, c1t;   // of output channels 4 (lane >> 4) .. + 3
    {
        const lz_c1_job &jb = a.c1[c1j];
#pragma unroll
        for (int g = 0; g < 4; ++g) c1w[g] = *reinterpret_cast<const float4 *>(jb.w + (size_t)(lane & 15) * 64 + g * 16 + (lane >> 4) * 4);
        c1b = *reinterpret_cast<const float4 *>(jb.bias + (lane >> 4) * 4);
        c1s = *reinterpret_cast<const float4 *>(jb.scale + (lane >> 4) * 4);
        c1t = *reinterpret_cast<const float4 *>(jb.shift + (lane >> 4) * 4);
    }
    int g_slot = 0, g_action = 0;
    if constexpr (TREE != 0) {
        int32_t *s_sel = reinterpret_cast<int32_t *>(sMisc + 120);
        // split heads: V is free until the first layer's input transform -- the leaf hand-over, the counters and the reduction
        // scratch of the head waves live there
        float *s_leaf = sV;
        int32_t *s_ctr = reinterpret_cast<int32_t *>(sV + 80);
        float *s_red = sV + 96;
        bool heads_on = false;
        if constexpr (HEADS) {
            static_assert(NW == 8, "seven head waves");
            heads_on = step.sh.on != 0;
            if (heads_on) {
                if (tid < 8) s_ctr[tid] = 0;
                __syncthreads();
            }
        }
        if (wv == 0) {
            // the tree step is one wave of strictly dependent instructions: it goes first wherever it competes with the head waves
            if constexpr (HEADS) __builtin_amdgcn_s_setprio(3);
            dev_step_lds<1, TREE - 1>(step.t, b, step.new_node, step.discount, step.vps, step.values, step.logits, step.horizon,
                                      step.a, step.delta, step.vtp, reinterpret_cast<float4 *>(smem), s_sel, step.ts,
                                      heads_on ? s_leaf : nullptr, s_ctr + 2, 3);
            if constexpr (HEADS) __builtin_amdgcn_s_setprio(0);
        } else {
            if constexpr (HEADS) {
                // head roles: waves 1-3 value, 5-7 value prefix, wave 4 -- which shares its SIMD with the tree wave -- the light policy head
                const int hw = wv < 4 ? wv - 1 : (wv == 4 ? 6 : wv - 2);
                if (heads_on) heads_in_prologue(step.sh, b, step.t.A, hw, lane, s_leaf, s_ctr, s_red, step.ts);
            }
            // the first layer's first weight fragments do not depend on the selection: the waves that are not the tree wave request
            // theirs now (7 / 8 of the 131 KB that used to be requested after the step -- ~3 k cycles of this launch), and stage the
            // folded-BatchNorm tables of all layers
            fill_ring();
            for (int i = tid - 64; i < a.nlayers * 128; i += NTHR - 64) {
                const int L = i >> 7, r = i & 127;
                sSS[i] = (r < 64) ? a.layer[L].scale[r] : a.layer[L].shift[r - 64];
            }
        }
        __syncthreads();
        if (step.ts && b == 0 && tid == 0) lz_stamp_store(step.ts + (5), __builtin_readcyclecounter());
        g_slot = s_sel[0];
        g_action = s_sel[1];
        if (wv == 0) fill_ring();
    } else {
        if (a.gather_ix) g_slot = a.gather_ix[b];
        if (a.act_table) g_action = a.action[b];
    }
    {
        const float *src = a.in + (size_t)b * HW * 64 + (size_t)g_slot * a.slot_stride;
        constexpr int NU = (HW * 16 + NTHR - 1) / NTHR;
        float4 v[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int idx = u * NTHR + tid;
            v[u] = vzero4();
#ifdef LZ_DEBUG_KNOBS
            if (a.debug_flags & 1) continue;
#endif
            if (idx < HW * 16) v[u] = *reinterpret_cast<const float4 *>(src + (size_t)idx * 4);
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int idx = u * NTHR + tid;
            if (idx < HW * 16) *reinterpret_cast<float4 *>(smem + (idx >> 4) * PS + (idx & 15) * 4) = v[u];
        }
        if (tid < 64) *reinterpret_cast<float4 *>(smem + (tid >> 4) * BUF + HW * PS + (tid & 15) * 4) = vzero4();
#ifdef LZ_DEBUG_KNOBS
        if (a.act_table && !BIG && !(a.debug_flags & 2)) {
#else
        if (a.act_table && !BIG) {
#endif
            const float *tsrc = a.act_table + (size_t)g_action * HW * 64;
            float4 tv[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int idx = min(u * NTHR + tid, HW * 16 - 1);
                tv[u] = *reinterpret_cast<const float4 *>(tsrc + (size_t)idx * 4);
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int idx = u * NTHR + tid;
                if (idx < HW * 16) *reinterpret_cast<float4 *>(sTab + (idx >> 4) * PS + (idx & 15) * 4) = tv[u];
            }
        }
        if (TREE == 0) {   // (with the tree step in the prologue the other waves staged these during the step)
            for (int i = tid; i < a.nlayers * 128; i += NTHR) {
                const int L = i >> 7, r = i & 127;
                sSS[i] = (r < 64) ? a.layer[L].scale[r] : a.layer[L].shift[r - 64];
            }
        }
    }
    // ---- per-thread geometry, the same for every layer
    // input transform item = (tile, channel quad[, row pair hr]): LDS offsets of the patch pixels it reads (outside the image:
    // the all-zero pixel).  V rows 2 hr, 2 hr + 1 need patch rows hr .. hr + 2 (KH = 2); all four rows otherwise.
    constexpr int PR = (KH == 2) ? 3 : 4;            // patch rows per item
    const int it = min(tid, NITEM - 1), it_hr = (KH == 2) ? it / (NT * 16) : 0, it_tile = (it / 16) % NT, it_cq = it & 15;
    int poff[PR * 4];
    {
        const int ty = it_tile / TW, tx = it_tile - ty * TW;
#pragma unroll
        for (int i = 0; i < PR; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int y = 2 * ty - 1 + it_hr + i, x = 2 * tx - 1 + j;
                const bool ok = (y >= 0) & (y < GH) & (x >= 0) & (x < GW);
                int o = (ok ? y * GW + x : HW) * PS + it_cq * 4;
                asm volatile("" : "+v"(o));
                poff[4 * i + j] = o;
            }
    }
    const int voff = it_tile * PS + it_cq * 4;
    // One LDS read serves all row blocks: lanes 4 rb .. 4 rb + 3 (block rb) hold tiles 4 rb .. 4 rb + 3 and the instruction's A
    // broadcast (CBSZ = 4: all 16 blocks take A from block ABID = rb) hands them to every block -- a third of the LDS traffic of
    // one read per row block (768 KB per layer through the LDS pipe was as long as the matrix work itself).
    const int aoff = ((4 * pi) * NT + min(lane & 15, NT - 1)) * PS + kh * KSW * 4;
    // combine step: this thread finishes output channel `lane` of the pixels u = wv + NW n (u = 4 tile + 2 dy + dx)
    int cpix[NOUT], cx[NOUT];
    float csg[NOUT];
#pragma unroll
    for (int n = 0; n < NOUT; ++n) {
        const int u = min(wv + NW * n, HW - 1), tile = u >> 2, dy = (u >> 1) & 1, dx = u & 1, ty = tile / TW, tx = tile - ty * TW;
        cpix[n] = ((2 * ty + dy) * GW + 2 * tx + dx) * PS + lane;
        cx[n] = ((dy * NT + tile) * 2 + dx) * 64 + lane;  // first of the three point rows this output sums: dy, dy + 1, dy + 2
        csg[n] = dy ? -1.0f : 1.0f;                        // Y0 = (X0 + X1) + X2, Y1 = (X1 - X2) - X3  (A^T = [1 1 1 0; 0 1 -1 -1])
    }
    const int kq4 = (lane >> 4) * 4, zoff = HW * PS;
    unsigned long long *sTS = reinterpret_cast<unsigned long long *>(sMisc);
    int nts = 0;
#define LZ_TS() do { if constexpr (TS) { if (b == 0 && tid == 0 && nts < 60) sTS[nts++] = __builtin_readcyclecounter(); } } while (0)
    LZ_TS();
    __syncthreads();
    LZ_TS();
    if constexpr (TREE != 0) { if (step.ts && b == 0 && tid == 0) lz_stamp_store(step.ts + (6), __builtin_readcyclecounter()); }

    for (int L = 0; L < a.nlayers; ++L) {
        const lz_chain_layer &ly = a.layer[L];
        const float *sIn = smem + ly.in * BUF;
        float *sOut = smem + ly.out * BUF;
        const bool more = L + 1 < a.nlayers;
        const f32x4 *wc = reinterpret_cast<const f32x4 *>(ly.uc) + wbase;
        const f32x4 *wn = reinterpret_cast<const f32x4 *>(a.layer[more ? L + 1 : L].uc) + wbase;
        // ---- input transform V = B^T d B
        if (tid < NITEM) {
            f32x4 d[PR * 4];
#pragma unroll
            for (int k = 0; k < PR * 4; ++k) d[k] = *reinterpret_cast<const f32x4 *>(sIn + poff[k]);
            if constexpr (TS) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); LZ_TS(); }
            if constexpr (KH == 1) {
                f32x4 e[16];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    e[0 + j] = pk_sub(d[0 + j], d[8 + j]);
                    e[4 + j] = d[4 + j] + d[8 + j];
                    e[8 + j] = pk_sub(d[8 + j], d[4 + j]);
                    e[12 + j] = pk_sub(d[4 + j], d[12 + j]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    *reinterpret_cast<f32x4 *>(sV + (4 * i + 0) * NT * PS + voff) = pk_sub(e[4 * i + 0], e[4 * i + 2]);
                    *reinterpret_cast<f32x4 *>(sV + (4 * i + 1) * NT * PS + voff) = e[4 * i + 1] + e[4 * i + 2];
                    *reinterpret_cast<f32x4 *>(sV + (4 * i + 2) * NT * PS + voff) = pk_sub(e[4 * i + 2], e[4 * i + 1]);
                    *reinterpret_cast<f32x4 *>(sV + (4 * i + 3) * NT * PS + voff) = pk_sub(e[4 * i + 1], e[4 * i + 3]);
                }
            } else {
                // row pair hr: patch rows r0 r1 r2 = hr, hr + 1, hr + 2.  hr = 0: e0 = r0 - r2, e1 = r1 + r2;  hr = 1: e2 = r1 - r0, e3 = r0 - r2
                f32x4 e[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 s02 = pk_sub(d[0 + j], d[8 + j]), a12 = d[4 + j] + d[8 + j], s10 = pk_sub(d[4 + j], d[0 + j]);
                    e[0 + j] = it_hr ? s10 : s02;
                    e[4 + j] = it_hr ? s02 : a12;
                }
                float *vb = sV + (it_hr * 8) * NT * PS + voff;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    *reinterpret_cast<f32x4 *>(vb + (4 * i + 0) * NT * PS) = pk_sub(e[4 * i + 0], e[4 * i + 2]);
                    *reinterpret_cast<f32x4 *>(vb + (4 * i + 1) * NT * PS) = e[4 * i + 1] + e[4 * i + 2];
                    *reinterpret_cast<f32x4 *>(vb + (4 * i + 2) * NT * PS) = pk_sub(e[4 * i + 2], e[4 * i + 1]);
                    *reinterpret_cast<f32x4 *>(vb + (4 * i + 3) * NT * PS) = pk_sub(e[4 * i + 1], e[4 * i + 3]);
                }
            }
        }
        if constexpr (TS) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); LZ_TS(); }
        if constexpr (TS) { LZ_TS(); }
        __syncthreads();
        LZ_TS();
        // ---- this wave's four points x its channel quads: NRB row blocks x 4 k per step
        f32x4 acc[4][NRB];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) acc[p][rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // A fragments are read AD steps (of 12 short MFMAs = 96 cycles) ahead: one step does not cover the LDS latency
        constexpr int AD = 3;
        f32x4 af[AD + 1];
        auto fetch_a = [&](int s, f32x4 &f) { f = *reinterpret_cast<const f32x4 *>(sV + aoff + (s / KSW) * NT * PS + (s % KSW) * 4); };
#pragma unroll
        for (int s = 0; s < AD; ++s) fetch_a(s, af[s]);
        // the epilogue's operands do not depend on the products and nobody writes them before the combine step: they are requested
        // behind the barrier and arrive under the products (in front of the barrier they were on the critical path of the waves that
        // carry a transform item)
        const float sc = sSS[L * 128 + lane], sh = sSS[L * 128 + 64 + lane];
        const bool tab = ly.act != 0, hasres = ly.res >= 0, relu = ly.relu != 0;
        const float *sRes = smem + max(ly.res, 0) * BUF;
        float tv[NOUT], rv[NOUT];
#pragma unroll
        for (int n = 0; n < NOUT; ++n) tv[n] = rv[n] = 0.0f;
        if (tab) {   // wave-uniform: only the dynamics convolution reads the action table, only a block's second layer a residual
#pragma unroll
            for (int n = 0; n < NOUT; ++n)
                tv[n] = BIG ? a.act_table[(size_t)g_action * HW * 64 + (cpix[n] / PS) * 64 + lane] : sTab[cpix[n]];
        }
        if (hasres) {
#pragma unroll
            for (int n = 0; n < NOUT; ++n) rv[n] = sRes[cpix[n]];
        }
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const f32x4 bfr = wq[s % R];
            wq[s % R] = (s + R < NSTEP) ? wc[wofs(s + R)] : wn[wofs(s + R - NSTEP)];
            if (s + AD < NSTEP) fetch_a(s + AD, af[(s + AD) % (AD + 1)]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                {   // the broadcast block id is an immediate
                    f32x4 (&ac)[NRB] = acc[s / KSW];
                    const float av = af[s % (AD + 1)][j], bv = bfr[j];
                    ac[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, ac[0], 4, 0, 0);
                    if constexpr (NRB > 1) ac[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, ac[1], 4, 1, 0);
                    if constexpr (NRB > 2) ac[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, ac[2], 4, 2, 0);
                    if constexpr (NRB > 3) ac[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, ac[3], 4, 3, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        LZ_TS();
        if constexpr (BIG) __syncthreads();   // the row results overwrite V: every wave is done reading it
        // ---- output transform, first half: point row pi times A (inside the lane), to LDS
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int tile = rb * 4 + q;
                if (tile < NT) {
                    const float m0 = acc[0][rb][q], m1 = acc[1][rb][q], m2 = acc[2][rb][q], m3 = acc[3][rb][q];
                    sX[((wv * NT + tile) * 2 + 0) * 64 + lane] = (m0 + m1) + m2;
                    sX[((wv * NT + tile) * 2 + 1) * 64 + lane] = (m1 - m2) - m3;
                }
            }
        if constexpr (TS) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); LZ_TS(); }
        __syncthreads();
        LZ_TS();
        // ---- second half + epilogue: BN (+ action table) (+ residual) (+ ReLU) -> LDS (and the latent pool)
        float outv[NOUT], xs[NOUT][3 * KH];
#pragma unroll
        for (int n = 0; n < NOUT; ++n)  // all reads first: the stores below may alias them as far as the compiler knows
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int h = 0; h < KH; ++h) xs[n][r * KH + h] = sX[cx[n] + (h * 4 + r) * NT * 128];
#pragma unroll
        for (int n = 0; n < NOUT; ++n) {
            float x0 = xs[n][0], x1 = xs[n][KH], x2 = xs[n][2 * KH];
            if constexpr (KH == 2) { x0 += xs[n][1]; x1 += xs[n][3]; x2 += xs[n][5]; }
            float v = (x0 + csg[n] * x1) + csg[n] * x2;
            v += tv[n];
            v = v * sc + sh;
            v += rv[n];
            if constexpr (GELU) v = ly.relu == 2 ? gelu_tanh_(v) : (relu ? fmaxf(v, 0.0f) : v);   // per-layer code (lz_chain_layer::relu)
            else v = relu ? fmaxf(v, 0.0f) : v;
            outv[n] = v;
            if (wv + NW * n < HW) sOut[cpix[n]] = v;
        }
        if (ly.gout) {
            float *go = ly.gout + (size_t)b * HW * 64;
#pragma unroll
            for (int n = 0; n < NOUT; ++n)
                if (wv + NW * n < HW) store_wt(go + (cpix[n] / PS) * 64 + lane, outv[n]);
        }
        if constexpr (TS) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); LZ_TS(); }
        __syncthreads();
        LZ_TS();
    }
    // 1x1 head convolutions (64 -> 16) + bias + BN + ReLU.  The weights are the A operand: D[channel][pixel] leaves a lane with
    // four consecutive channels of one pixel, so a tile goes out as one coalesced 1 KB store (the transposed product took 12
    // scattered 4-byte stores per lane: 2 k cycles of the kernel's tail)
    auto c1_store = [&](const lz_c1_job &jb, int row, int cq, const f32x4 &acc) {   // c1b / c1s / c1t are those of channel quad cq
        float4 v;
        v.x = (acc[0] + c1b.x) * c1s.x + c1t.x;
        v.y = (acc[1] + c1b.y) * c1s.y + c1t.y;
        v.z = (acc[2] + c1b.z) * c1s.z + c1t.z;
        v.w = (acc[3] + c1b.w) * c1s.w + c1t.w;
        bool g = false;
        if constexpr (GELU) g = jb.act == 2;   // per-job code (lz_c1_job::act)
        if (g) { v.x = gelu_tanh_(v.x); v.y = gelu_tanh_(v.y); v.z = gelu_tanh_(v.z); v.w = gelu_tanh_(v.w); }
        else { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
        store_wt(jb.out + ((size_t)b * HW + row) * jb.out_stride + jb.out_off + cq * 4, (f32x4){v.x, v.y, v.z, v.w});
    };
    auto c1_tile = [&](int job, int i) {   // 16 pixels from 16 i
        const float *sIn = smem + a.c1_in[job] * BUF + kq4;
        const int row = i * 16 + (lane & 15);
        const int off = (row < HW) ? row * PS : zoff;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 xf = *reinterpret_cast<const float4 *>(sIn + off + g * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vget(c1w[g], j), vget(xf, j), acc, 0, 0, 0);
        }
        if (row < HW) c1_store(a.c1[job], row, lane >> 4, acc);
    };
    // remainder pixels 32 .. 35: 16 blocks of (4 channels) x (4 pixels), block = (k quarter, channel quad) -- the lane layout the weight
    // fragment already has; the four k-quarter partial sums are added across lanes (every quarter then holds the total), and of the
    // four lanes that hold (quad cg, pixel j) the one whose k quarter equals cg stores: its epilogue operands are the right quad's
    auto c1_rem = [&](int job) {
        const float *sIn = smem + a.c1_in[job] * BUF + kq4;
        const int row = (HW / 16) * 16 + (lane & 3);
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 xf = *reinterpret_cast<const float4 *>(sIn + row * PS + g * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(vget(c1w[g], j), vget(xf, j), acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = xor32_sum(xor16_sum(acc[q]));
        const int cg = (lane >> 2) & 3;
        if ((lane >> 4) == cg) c1_store(a.c1[job], row, cg, acc);
    };
    if constexpr (C1SPLIT) {
        if (wv < a.nc1) { c1_tile(c1j, 0); c1_rem(c1j); }
        else if (wv < 2 * a.nc1) c1_tile(c1j, 1);
    } else if (wv < a.nc1) {
#pragma unroll
        for (int i = 0; i < MT; ++i) c1_tile(wv, i);
    }
    LZ_TS();
    if constexpr (TS) {
        if (b == 0 && tid == 0 && a.tstamp) {
            a.tstamp[0] = (unsigned long long)nts;
            for (int i = 0; i < nts && i < 60; ++i) a.tstamp[1 + i] = sTS[i];
        }
    }
#undef LZ_TS
    lz_stamp_end(a.stamp);
}

// ------------------------------------------------------------------------------------------------
// FAST MODE (lz_model_cfg::precision = 1; BASELINE.md section 2, arm "fast mode": reported separately, statistical parity only).
// The same chain as k_chain_w -- same arguments, same prologue (tree step on wave 0, split heads on waves 1-7), same 1x1 head convolutions
// -- with the 3x3 convolutions in the direct form on v_mfma_f32_16x16x32_bf16: weights and the A operand (activations) rounded to bf16
// (RNE), fp32 accumulation, fp32 BatchNorm / action table / residual / ReLU, fp32 latents in the pool.  The matrix work shrinks to 27
// MFMAs per wave and layer, so no Winograd transform (whose weights are 1.78x the bytes): a layer streams 74 KB of weights per
// workgroup instead of 262 KB.  Work split: wave (nt = w & 3, kh = w >> 2) owns the 16-channel output tile nt for ALL pixels (three
// 16-pixel row tiles, 36 rows used) over input channels 32 kh .. 32 kh + 31 of every tap: its 9 weight fragments of a layer (one per tap,
// 16 B per lane) sit in registers and the next layer's are requested at the layer's start; the two k halves meet in LDS.  Activations live in
// LDS twice: fp32 (residual, action table add, head convs, pool) and a bf16 copy [pixel][80] that serves the pixel operand with ONE
// conflict-free ds_read_b128 per (pixel tile, tap).
// ------------------------------------------------------------------------------------------------
template <int GW, int GH, int TREE = 0, bool HEADS = false>
__global__ __launch_bounds__(512) void k_chain_b(lz_chain_args a, typename step_arg<TREE>::type step)
{
    constexpr int NW = 8, PS = 68, HW = GW * GH, MT = (HW + 15) / 16, BUF = (HW + 1) * PS, NTHR = NW * 64;
    constexpr int PB = 80;                               // bf16 per pixel of the bf16 copies: 64 + pad.  160 B = 10 bank quads: the 16-lane groups ds_read_b128 is
                                                         // served in ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS) then hit 16 distinct quads (144 B: 2-way conflicts)
    static_assert(MT <= 4 && HW % 4 == 0 && (!HEADS || MT == 3), "up to 64 pixels; split heads on the 6x6 latent only");
    extern __shared__ __attribute__((aligned(16))) float smem[];  // 4 fp32 activation buffers of BUF floats (the staged tree first), then
    float *sTab = smem + 4 * BUF;                       // [HW][PS] one-hot-action table slice of this root's action
    float *sSS = sTab + HW * PS;                        // [LZ_CHAIN_MAX_LAYERS][2][64] folded-BN scale / shift
    float *sMisc = sSS + LZ_CHAIN_MAX_LAYERS * 128;     // 128 floats: the tree step's selection
    float *sP = sMisc + 128;                            // [4 nt][MT][64 lanes][4] partial sums of the kh = 1 waves (prologue: head scratch)
    __bf16 *sB = reinterpret_cast<__bf16 *>(sP + 4 * MT * 256);   // 4 x [HW + 1][PB] bf16 copies of the activation buffers
    constexpr int BB = (HW + 1) * PB;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nt = wv & 3, kh = wv >> 2;
    const int b = blockIdx.x;
    lz_stamp_begin(a.stamp);
    // per-layer parameters: lane L keeps layer L's (a scalar load from the argument block at the top of every layer is a round trip the
    // layer then waits for); v_readlane hands them out
    const lz_chain_layer &myl = a.layer[min(lane, LZ_CHAIN_MAX_LAYERS - 1)];
    const unsigned long long my_wb = (unsigned long long)myl.wb, my_gout = (unsigned long long)myl.gout;
    const int my_flags = myl.in | (myl.out << 2) | ((myl.res + 1) << 4) | ((myl.relu != 0) << 7) | ((myl.act != 0) << 8);
    auto lane64 = [&](unsigned long long v, int L) {
        const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, L), hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), L);
        return ((unsigned long long)hi << 32) | lo;
    };
    // weights: [layer][kh][nt][tap][64 lanes][8 bf16]
    const size_t wofs = (((size_t)kh * 4 + nt) * 9) * 64 + lane;
    bf16x8 wc[9];
    auto load_w0 = [&]() {
        const bf16x8 *w0 = reinterpret_cast<const bf16x8 *>(a.layer[0].wb) + wofs;
#pragma unroll
        for (int t = 0; t < 9; ++t) wc[t] = w0[t * 64];
    };
    // the first layer's weights: at once without a tree step; with one, the head waves request theirs when their head is done and the tree
    // wave after the step, together with the latent gather (36 more live registers across the prologue spill)
    if constexpr (TREE == 0) load_w0();
    // 1x1 head convolutions at the end of the kernel (as in k_chain_w)
    constexpr bool C1SPLIT = (HW % 16) != 0 && (HW % 16) <= 4 && HW / 16 == 2;
    const int nj = max(a.nc1, 1);
    const int c1j = C1SPLIT ? wv % nj : min(wv, nj - 1);
    float4 c1w[4];
    float4 c1b, c1s, c1t;
    {
        const lz_c1_job &jb = a.c1[c1j];
#pragma unroll
        for (int g = 0; g < 4; ++g) c1w[g] = *reinterpret_cast<const float4 *>(jb.w + (size_t)(lane & 15) * 64 + g * 16 + (lane >> 4) * 4);
        c1b = *reinterpret_cast<const float4 *>(jb.bias + (lane >> 4) * 4);
        c1s = *reinterpret_cast<const float4 *>(jb.scale + (lane >> 4) * 4);
        c1t = *reinterpret_cast<const float4 *>(jb.shift + (lane >> 4) * 4);
    }
    int g_slot = 0, g_action = 0;
    if constexpr (TREE != 0) {
        int32_t *s_sel = reinterpret_cast<int32_t *>(sMisc + 120);
        float *s_leaf = sP;
        int32_t *s_ctr = reinterpret_cast<int32_t *>(sP + 80);
        float *s_red = sP + 96;
        bool heads_on = false;
        if constexpr (HEADS) {
            heads_on = step.sh.on != 0;
            if (heads_on) {
                if (tid < 8) s_ctr[tid] = 0;
                __syncthreads();
            }
        }
        if (wv == 0) {
            if constexpr (HEADS) __builtin_amdgcn_s_setprio(3);
            dev_step_lds<1, TREE - 1>(step.t, b, step.new_node, step.discount, step.vps, step.values, step.logits, step.horizon,
                                      step.a, step.delta, step.vtp, reinterpret_cast<float4 *>(smem), s_sel, step.ts,
                                      heads_on ? s_leaf : nullptr, s_ctr + 2, 3);
            if constexpr (HEADS) __builtin_amdgcn_s_setprio(0);
        } else {
            if constexpr (HEADS) {
                const int hw = wv < 4 ? wv - 1 : (wv == 4 ? 6 : wv - 2);
                if (heads_on) heads_in_prologue(step.sh, b, step.t.A, hw, lane, s_leaf, s_ctr, s_red, step.ts);
            }
            load_w0();
            for (int i = tid - 64; i < a.nlayers * 128; i += NTHR - 64) {
                const int L = i >> 7, r = i & 127;
                sSS[i] = (r < 64) ? a.layer[L].scale[r] : a.layer[L].shift[r - 64];
            }
        }
        __syncthreads();
        if (step.ts && b == 0 && tid == 0) lz_stamp_store(step.ts + (5), __builtin_readcyclecounter());
        g_slot = s_sel[0];
        g_action = s_sel[1];
        if (wv == 0) load_w0();
    } else {
        if (a.gather_ix) g_slot = a.gather_ix[b];
        if (a.act_table) g_action = a.action[b];
    }
    {
        const float *src = a.in + (size_t)b * HW * 64 + (size_t)g_slot * a.slot_stride;
        constexpr int NU = (HW * 16 + NTHR - 1) / NTHR;
        float4 v[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int idx = min(u * NTHR + tid, HW * 16 - 1);
            v[u] = *reinterpret_cast<const float4 *>(src + (size_t)idx * 4);
        }
        float4 tv[NU];
        const bool tabl = a.act_table != nullptr;
        {
            const float *tsrc = tabl ? a.act_table + (size_t)g_action * HW * 64 : src;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int idx = min(u * NTHR + tid, HW * 16 - 1);
                tv[u] = *reinterpret_cast<const float4 *>(tsrc + (size_t)idx * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int idx = u * NTHR + tid;
            if (idx < HW * 16) {
                *reinterpret_cast<float4 *>(smem + (idx >> 4) * PS + (idx & 15) * 4) = v[u];
                bf16x4 h;
                h[0] = (__bf16)v[u].x; h[1] = (__bf16)v[u].y; h[2] = (__bf16)v[u].z; h[3] = (__bf16)v[u].w;
                *reinterpret_cast<bf16x4 *>(sB + (idx >> 4) * PB + (idx & 15) * 4) = h;
                if (tabl) *reinterpret_cast<float4 *>(sTab + (idx >> 4) * PS + (idx & 15) * 4) = tv[u];
            }
        }
        // the all-zero pixel of every buffer (fp32: the head convolutions' padding rows; bf16: the halo)
        if (tid < 64) *reinterpret_cast<float4 *>(smem + (tid >> 4) * BUF + HW * PS + (tid & 15) * 4) = vzero4();
        if (tid >= 64 && tid < 64 + 4 * (PB / 8)) {
            const int i = tid - 64;
            *reinterpret_cast<float4 *>(sB + (i / (PB / 8)) * BB + HW * PB + (i % (PB / 8)) * 8) = vzero4();
        }
        if (TREE == 0) {
            for (int i = tid; i < a.nlayers * 128; i += NTHR) {
                const int L = i >> 7, r = i & 127;
                sSS[i] = (r < 64) ? a.layer[L].scale[r] : a.layer[L].shift[r - 64];
            }
        }
    }
    // ---- per-lane geometry, the same for every layer: A rows of this lane = pixels 16 mt + (lane & 15); tap (dy, dx) reads pixel
    // m + dy GW + dx when it is inside the image, the zero pixel otherwise (one validity bit per (row tile, tap))
    unsigned long long valid = 0;
    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + (lane & 15), y = m / GW, x = m - y * GW;
        abase[mt] = (m * PB + kh * 32 + (lane >> 4) * 8) * 2;   // bytes
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            if (m < HW && yy >= 0 && yy < GH && xx >= 0 && xx < GW) valid |= 1ull << (mt * 9 + t);
        }
    }
    const int azero = (HW * PB + kh * 32 + (lane >> 4) * 8) * 2;
    const int kq4 = (lane >> 4) * 4, zoff = HW * PS;
    __syncthreads();
    if constexpr (TREE != 0) { if (step.ts && b == 0 && tid == 0) lz_stamp_store(step.ts + (6), __builtin_readcyclecounter()); }

    // output geometry of this lane, the same for every layer.  The MFMAs run TRANSPOSED (weights as the A operand): D[channel][pixel], so a
    // lane ends up with four consecutive channels co4 .. co4 + 3 of ONE pixel -- 16 contiguous bytes in every [pixel][channel] array
    const int co4 = nt * 16 + 4 * (lane >> 4);
    const int nlayers = a.nlayers;
    for (int L = 0; L < nlayers; ++L) {
        const int flags = __builtin_amdgcn_readlane(my_flags, L);
        const int Ln = L + 1 < nlayers ? L + 1 : L;
        const char *sBin = reinterpret_cast<const char *>(sB + (flags & 3) * BB);
        float *sOut = smem + ((flags >> 2) & 3) * BUF;
        __bf16 *sBout = sB + ((flags >> 2) & 3) * BB;
        // the next layer's weights first: they have this whole layer to arrive (left to the scheduler the requests sink behind the
        // MFMAs and every layer starts by waiting for its weights: 2.8 us per layer, measured)
        bf16x8 wn[9];
        {
            gbl_bf16x8 *w1 = as_global_bf16x8(lane64(my_wb, Ln)) + wofs;
#ifdef LZ_DEBUG_KNOBS
            if (a.debug_flags & 16) {   // 16 = no weight stream (the registers keep the first layer's)
#pragma unroll
                for (int t = 0; t < 9; ++t) wn[t] = wc[t];
            } else
#endif
#pragma unroll
            for (int t = 0; t < 9; ++t) wn[t] = w1[t * 64];
        }
        // pixel fragments by kernel row (3 taps x MT row tiles = one group): two groups are requested before the first MFMA, the third
        // goes into the first group's registers behind its MFMAs -- one LDS latency per layer instead of one per fragment
        auto read_group = [&](int g, bf16x8 (&af)[3][MT]) {
#pragma unroll
            for (int tt = 0; tt < 3; ++tt) {
                const int t = 3 * g + tt;
                const int toff = ((t / 3 - 1) * GW + (t % 3 - 1)) * PB * 2;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int off = (valid >> (mt * 9 + t)) & 1 ? abase[mt] + toff : azero;
                    af[tt][mt] = *reinterpret_cast<const bf16x8 *>(sBin + off);
                }
            }
        };
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto mma_group = [&](int g, const bf16x8 (&af)[3][MT]) {
#pragma unroll
            for (int tt = 0; tt < 3; ++tt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[3 * g + tt], af[tt][mt], acc[mt], 0, 0, 0);
        };
        if constexpr (MT == 4) {
            // 8x8 latent: four pixel tiles per tap; a three-tap-deep ring (two 12-fragment groups do not fit the register file beside the weights)
            auto read_tap = [&](int t, bf16x8 (&af)[MT]) {
                const int toff = ((t / 3 - 1) * GW + (t % 3 - 1)) * PB * 2;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int off = (valid >> (mt * 9 + t)) & 1 ? abase[mt] + toff : azero;
                    af[mt] = *reinterpret_cast<const bf16x8 *>(sBin + off);
                }
            };
            bf16x8 ring[3][MT];
            read_tap(0, ring[0]);
            read_tap(1, ring[1]);
            read_tap(2, ring[2]);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[t], ring[t % 3][mt], acc[mt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (t + 3 < 9) read_tap(t + 3, ring[t % 3]);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
        bf16x8 a0[3][MT], a1[3][MT];
#ifdef LZ_DEBUG_KNOBS
        if (!(a.debug_flags & 4)) {   // timing experiments (debug build; results are then wrong): 4 = no pixel reads / MFMAs
#endif
        read_group(0, a0);
        read_group(1, a1);
        __builtin_amdgcn_sched_barrier(0);
        mma_group(0, a0);
        __builtin_amdgcn_sched_barrier(0);
        read_group(2, a0);
        __builtin_amdgcn_sched_barrier(0);
        mma_group(1, a1);
        __builtin_amdgcn_sched_barrier(0);
        mma_group(2, a0);
        __builtin_amdgcn_sched_barrier(0);
#ifdef LZ_DEBUG_KNOBS
        }
#endif
        }
        // ---- the two k halves meet in LDS: the kh = 0 wave of an output tile finishes the even pixel tiles (6x6: pixels 0..15, 32..35), the
        // kh = 1 wave the odd ones -- each leaves its partial sums of the OTHER wave's tiles in sP[nt][mt]
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            if ((mt & 1) != kh) *reinterpret_cast<f32x4 *>(sP + ((nt * MT + mt) * 64 + lane) * 4) = acc[mt];
        __syncthreads();
#ifdef LZ_DEBUG_KNOBS
        if (!(a.debug_flags & 8))     // 8 = no epilogue
#endif
        {
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(sSS + L * 128 + co4), sh = *reinterpret_cast<const f32x4 *>(sSS + L * 128 + 64 + co4);
            const bool tab = (flags >> 8) & 1, relu = (flags >> 7) & 1;
            const int res = ((flags >> 4) & 7) - 1;
            const float *sRes = smem + max(res, 0) * BUF;
            float *go = reinterpret_cast<float *>(lane64(my_gout, L));
            if (go) go += (size_t)b * HW * 64;
            // every LDS read of the epilogue before its first write (the compiler must assume they alias)
            constexpr int NF = (MT + 1) / 2;       // pixel tiles this wave may finish: kh, kh + 2 (< MT)
            f32x4 other[NF], tvv[NF], rvv[NF];
            int mpix[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int mt = min(kh + 2 * f, MT - 1);
                mpix[f] = mt * 16 + (lane & 15);
                const int m = min(mpix[f], HW - 1);
                other[f] = *reinterpret_cast<const f32x4 *>(sP + ((nt * MT + mt) * 64 + lane) * 4);
                tvv[f] = *reinterpret_cast<const f32x4 *>(sTab + m * PS + co4);
                rvv[f] = *reinterpret_cast<const f32x4 *>(sRes + m * PS + co4);
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (kh + 2 * f >= MT) continue;
                const f32x4 mine = kh == 1 ? acc[(1 + 2 * f < MT) ? 1 + 2 * f : MT - 1] : acc[2 * f];
                f32x4 o;
                bf16x4 ob;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = mine[q] + other[f][q];
                    v += tab ? tvv[f][q] : 0.0f;
                    v = v * sc[q] + sh[q];
                    v += res >= 0 ? rvv[f][q] : 0.0f;
                    o[q] = relu ? fmaxf(v, 0.0f) : v;
                    ob[q] = (__bf16)o[q];
                }
                if (mpix[f] < HW) {
                    *reinterpret_cast<f32x4 *>(sOut + mpix[f] * PS + co4) = o;
                    *reinterpret_cast<bf16x4 *>(sBout + mpix[f] * PB + co4) = ob;
                    if (go) store_wt(go + mpix[f] * 64 + co4, o);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) wc[t] = wn[t];
        __syncthreads();
        if constexpr (TREE != 0) { if (step.ts && b == 0 && tid == 0 && L < 8) lz_stamp_store(step.ts + (16 + L), __builtin_readcyclecounter()); }
    }
    // 1x1 head convolutions (64 -> 16) + bias + BN + ReLU in fp32, as in k_chain_w
    auto c1_store = [&](const lz_c1_job &jb, int row, int cq, const f32x4 &acc) {
        float4 v;
        v.x = fmaxf((acc[0] + c1b.x) * c1s.x + c1t.x, 0.0f);
        v.y = fmaxf((acc[1] + c1b.y) * c1s.y + c1t.y, 0.0f);
        v.z = fmaxf((acc[2] + c1b.z) * c1s.z + c1t.z, 0.0f);
        v.w = fmaxf((acc[3] + c1b.w) * c1s.w + c1t.w, 0.0f);
        store_wt(jb.out + ((size_t)b * HW + row) * jb.out_stride + jb.out_off + cq * 4, (f32x4){v.x, v.y, v.z, v.w});
    };
    auto c1_tile = [&](int job, int i) {
        const float *sIn = smem + a.c1_in[job] * BUF + kq4;
        const int row = i * 16 + (lane & 15);
        const int off = (row < HW) ? row * PS : zoff;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 xf = *reinterpret_cast<const float4 *>(sIn + off + g * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vget(c1w[g], j), vget(xf, j), acc, 0, 0, 0);
        }
        if (row < HW) c1_store(a.c1[job], row, lane >> 4, acc);
    };
    auto c1_rem = [&](int job) {
        const float *sIn = smem + a.c1_in[job] * BUF + kq4;
        const int row = (HW / 16) * 16 + (lane & 3);
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 xf = *reinterpret_cast<const float4 *>(sIn + row * PS + g * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(vget(c1w[g], j), vget(xf, j), acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = xor32_sum(xor16_sum(acc[q]));
        const int cg = (lane >> 2) & 3;
        if ((lane >> 4) == cg) c1_store(a.c1[job], row, cg, acc);
    };
    if constexpr (C1SPLIT) {
        if (wv < a.nc1) { c1_tile(c1j, 0); c1_rem(c1j); }
        else if (wv < 2 * a.nc1) c1_tile(c1j, 1);
    } else if (wv < a.nc1) {
#pragma unroll
        for (int i = 0; i < MT; ++i) c1_tile(wv, i);
    }
    if constexpr (TREE != 0) { if (step.ts && b == 0 && tid == 0) lz_stamp_store(step.ts + (24), __builtin_readcyclecounter()); }
    lz_stamp_end(a.stamp);
}


// ------------------------------------------------------------------------------------------------
// PARITY MODE (fp32 accuracy), round 5: the recurrent chain on the 6x6 latent as SPLIT-bf16 products -- k_chain_b's launch (tree step on
// wave 0, split-head finish on waves 1-7, five 3x3 layers in the DIRECT form with the activations resident in LDS, 1x1 head convolutions)
// with every operand split EXACTLY into three bf16 terms and six of the nine cross products per k-step (see k_conv_s3 for the arithmetic:
// the error of a dot product is that of an fp32 FMA chain; tests/test_nn_gpu.py holds every simulation of a 256 x 50 search to 1e-5 (1 + |x|)
// of the torch fp32 modules; tests/test_split_bf16_cpu.py restates the arithmetic in numpy).  Why it is faster than the fp32 Winograd
// chain (k_chain_w): three bf16 planes of the direct form are 221 KB per layer against 262 KB of transformed fp32 weights, the six products
// are 5.2 k matrix cycles per SIMD and layer against 6.1 k, and the Winograd input / output transforms -- 3.7 k cycles of LDS-bound work
// per layer, four barriers -- are gone; a layer is 8.05 k cycles (products 5.7 k, at 91 % of the matrix rate) against 11.3 k.  Wave roles,
// the schedule of a tap and what the ISA showed on the way: DESIGN.md 3.2c.
// ------------------------------------------------------------------------------------------------
// The launch's body as a device function: k_chain_s3 is the launch-per-simulation form (b = blockIdx.x), k_search_resident (below) calls it
// once per simulation with RES = true -- `rs` then says which simulation of the resident launch this is (the per-simulation pointers are
// linear in it), loads of data ANOTHER workgroup produced inside the launch (the LSTM phase's head partials) bypass this CU's L1 (sc1:
// served by the XCD's L2, which the groups of 16 roots never leave), and the rows the LSTM phase reads leave as plain stores (they stay in
// that L2; write-through sc1 stores drop the line).
// RES: 0 the launch-per-simulation kernel | 1 k_sim_fused (only the head partials come from other workgroups of the launch: sc1 loads; the
// outputs are read by LATER launches and stay write-through) | 2 a launch that loops over simulations (per-simulation offsets `rs`, plain
// stores for what its own later phases read; not shipped: see k_sim_fused)
// LDS layout of the 6x6 split-bf16 chain (chain_s3_body; also k_sim_fused and the launchers): the fp32 buffers, scale / shift, selection words and the
// partial-sum area as before; the bf16 planes start on a 256-byte boundary and every plane is padded to a multiple of 256 bytes whose LAST 256 bytes
// are zero -- the "zero pixel" of an out-of-image tap is read from there at the 16-byte slot (bank quad) the lane's in-image read would have
// used, so that a 16-lane group of ds_read_b128 stays on 16 distinct quads whatever mix of in-image and zero reads it holds (one shared zero pixel
// collided with the lane that owned its quad: 1.33 M SQ_LDS_BANK_CONFLICT per launch).
constexpr int LZ_S3_HW = 36, LZ_S3_PB = 80;
constexpr size_t LZ_S3_SB_OFF = (((size_t)(4 * (LZ_S3_HW + 1) * 68 + LZ_CHAIN_MAX_LAYERS * 128 + 128 + 2 * 6 * 3 * 256) * 4 + 255) / 256) * 256;   // bytes
constexpr size_t LZ_S3_PLANE = ((((size_t)(LZ_S3_HW + 1) * LZ_S3_PB * 2 + 255) / 256) * 256);   // bytes per plane: 6144
static_assert(LZ_S3_PLANE - (size_t)LZ_S3_HW * LZ_S3_PB * 2 >= 256, "room for the zero line behind the pixels");
constexpr size_t LZ_S3_LDS = LZ_S3_SB_OFF + 4 * 3 * LZ_S3_PLANE;   // 4 buffers x 3 planes
template <int GW, int GH, int TREE, bool HEADS, int RES>
__device__ __forceinline__ void chain_s3_body(const lz_chain_args &a, const typename step_arg<TREE>::type &step, const int b, const int nroots, const lz_res_sim &rs)
{
    constexpr int NW = 8, PS = 68, HW = GW * GH, MT = (HW + 15) / 16, BUF = (HW + 1) * PS, NTHR = NW * 64;
    constexpr int PB = 80;                               // bf16 per pixel of the bf16 planes: 64 + pad.  160 B = 10 bank quads: the 16-lane groups ds_read_b128 is
                                                         // served in ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS) then hit 16 distinct quads (144 B: 2-way conflicts)
    static_assert(MT == 3 && HW % 4 == 0, "the 6x6 latent (three 16-pixel tiles)");
    constexpr int NPL = 3;                               // bf16 planes of every activation / weight: hi | mid | lo (exact three-term split of fp32)
    extern __shared__ __attribute__((aligned(16))) float smem[];  // 4 fp32 activation buffers of BUF floats (the staged tree first), then
    float *sSS = smem + 4 * BUF;                        // [LZ_CHAIN_MAX_LAYERS][2][64] folded-BN scale / shift
    float *sMisc = sSS + LZ_CHAIN_MAX_LAYERS * 128;     // 128 floats: the tree step's selection
    float *sP = sMisc + 128;                            // [2 tile pairs][6 tiles][3 other waves][64 lanes][4] partial sums (prologue: head scratch)
    static_assert(HW == LZ_S3_HW && PB == LZ_S3_PB, "the layout constants above");
    __bf16 *sB = reinterpret_cast<__bf16 *>(reinterpret_cast<char *>(smem) + LZ_S3_SB_OFF);   // 4 buffers x 3 planes x ([HW][PB] pixels, zeros up to the plane's end)
    constexpr int BB = (int)(LZ_S3_PLANE / 2);          // one plane (bf16 elements)
    constexpr int ZLINE = (int)LZ_S3_PLANE - 256;       // byte offset of the plane's zero line
    constexpr int BB3 = NPL * BB;                       // one buffer
    // Wave roles.  With one 16-channel output tile per wave every pixel fragment was read by four waves (648 KB of ds_read_b128 per layer and CU, 125 B/clk
    // at the matrix rate) and the LDS read path bound the launch.  (Round 6, with the layout below and the zero line: the product phase is 91 % matrix work.)
    // So a wave owns TWO output-channel tiles (np: tiles 2 np, 2 np + 1) for all three pixel tiles, one half of the input channels (kh) and one
    // half of the taps (th = 0: taps 0-4, th = 1: taps 5-8; waves w and w + 4 share a SIMD, so every SIMD gets 5 + 4 taps): each pixel fragment
    // feeds 12 MFMAs instead of 6, the LDS read traffic halves.  Four waves hold partial sums of the same six output tiles; they meet in LDS.
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, np = wv & 1, kh = (wv >> 1) & 1, th = wv >> 2, widx = wv >> 1;   // widx = kh + 2 th
    // per-layer parameters: lane L keeps layer L's (a scalar load from the argument block at the top of every layer is a round trip the
    // layer then waits for); v_readlane hands them out
    const lz_chain_layer &myl = a.layer[min(lane, LZ_CHAIN_MAX_LAYERS - 1)];
    const unsigned long long my_wb = (unsigned long long)myl.w3, my_gout = (unsigned long long)myl.gout;
    const int my_flags = myl.in | (myl.out << 2) | ((myl.res + 1) << 4) | ((myl.relu != 0) << 7) | ((myl.act != 0) << 8);
    auto lane64 = [&](unsigned long long v, int L) {
        const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, L), hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), L);
        return ((unsigned long long)hi << 32) | lo;
    };
    // weights: [layer][kh][nt][tap][plane][64 lanes][8 bf16].  A wave's fragments of a layer (2 tiles x 5 | 4 taps x 3 planes) stream through
    // a ring of RT taps: a slot is refilled right behind the products that read it -- with the same layer's tap RT further on, or, for the
    // last RT taps, with the NEXT layer's first taps (so the stream runs through the partial-sum exchange and the epilogue)
    constexpr int RT = 2;
    const int t0 = th ? 5 : 0;
    const size_t wofs0 = (((size_t)kh * 4 + 2 * np) * 9) * NPL * 64 + lane, wofs1 = wofs0 + (size_t)9 * NPL * 64;
    bf16x8 wr[RT][2][NPL];
    auto load_w = [&](gbl_bf16x8 *wl, int t, int pl, bf16x8 (&dst)[2][NPL]) {
        gbl_bf16x8 *w0 = wl + (size_t)(t * NPL + pl) * 64;
        dst[0][pl] = w0[wofs0];
        dst[1][pl] = w0[wofs1];
    };
    auto load_tap = [&](gbl_bf16x8 *wl, int t, bf16x8 (&dst)[2][NPL]) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) load_w(wl, t, pl, dst);
    };
    auto load_w0 = [&]() {
#pragma unroll
        for (int i = 0; i < RT; ++i) load_tap(as_global_bf16x8((unsigned long long)a.layer[0].w3), t0 + i, wr[i]);
    };
    // L2 does not keep anything across a kernel boundary, so the first workgroup of an XCD to ask for a weight line waits for the memory side
    // (Infinity Cache), and with the 32 workgroups of an XCD running in step all of them wait with it: the stream then runs at the miss
    // latency.  The workgroups of an XCD (blockIdx % 8: the dispatcher deals them round-robin) therefore touch one 32nd of every layer's
    // lines each while the tree step runs -- one load instruction per layer and workgroup, results unused -- and the layers find them in L2.
    auto prefetch_weights = [&](int L) {
        if (L >= a.nlayers) return;
        const int nr = min(max(nroots >> 3, 1), 32), r = (b >> 3) % nr;
        constexpr int LINES = 2 * 4 * 9 * NPL * 64 * 16 / 128;       // 128-byte lines of a layer
        const char *w = reinterpret_cast<const char *>(a.layer[L].w3);
        for (int ln = r + lane * nr; ln < LINES; ln += 64 * nr) (void)*reinterpret_cast<const volatile int *>(w + (size_t)ln * 128);
    };
    // the first layer's weights: at once without a tree step; with one, the head waves request theirs when their head is done and the tree
    // wave after the step, together with the latent gather
    if constexpr (TREE == 0) { load_w0(); prefetch_weights(wv); prefetch_weights(wv + 8); }
    // 1x1 head convolutions at the end of the kernel (as in k_chain_w)
    constexpr bool C1SPLIT = (HW % 16) != 0 && (HW % 16) <= 4 && HW / 16 == 2;
    const int nj = max(a.nc1, 1);
    const int c1j = C1SPLIT ? wv % nj : min(wv, nj - 1);
    float4 c1w[4];
    float4 c1b, c1s, c1t;
    {
        const lz_c1_job &jb = a.c1[c1j];
#pragma unroll
        for (int g = 0; g < 4; ++g) c1w[g] = *reinterpret_cast<const float4 *>(jb.w + (size_t)(lane & 15) * 64 + g * 16 + (lane >> 4) * 4);
        c1b = *reinterpret_cast<const float4 *>(jb.bias + (lane >> 4) * 4);
        c1s = *reinterpret_cast<const float4 *>(jb.scale + (lane >> 4) * 4);
        c1t = *reinterpret_cast<const float4 *>(jb.shift + (lane >> 4) * 4);
    }
    int g_slot = 0, g_action = 0;
    if constexpr (TREE != 0) {
        int32_t *s_sel = reinterpret_cast<int32_t *>(sMisc + 120);
        float *s_leaf = sP;
        int32_t *s_ctr = reinterpret_cast<int32_t *>(sP + 80);
        float *s_red = sP + 96;
        bool heads_on = false;
        if constexpr (HEADS) {
            heads_on = step.sh.on != 0;
            if (heads_on) {
                if (tid < 8) s_ctr[tid] = 0;
                __syncthreads();
            }
        }
        if (wv == 0) {
            if constexpr (HEADS) __builtin_amdgcn_s_setprio(3);
            if constexpr (RES == 2) {   // simulation rs.ds of the resident launch: the leaf's slot, its output rows and the draw counter move with it
                lz_traverse_args ta = step.a;
                ta.counter += (uint32_t)rs.ds;
                dev_step_lds<1, TREE - 1>(step.t, b, step.new_node + rs.ds, step.discount, step.vps + (size_t)rs.ds * rs.B, step.values + (size_t)rs.ds * rs.B,
                                          step.logits + (size_t)rs.ds * rs.BA, step.horizon, ta, step.delta, step.vtp, reinterpret_cast<float4 *>(smem), s_sel, step.ts,
                                          heads_on ? s_leaf : nullptr, s_ctr + 2, 3);
            } else
            dev_step_lds<1, TREE - 1>(step.t, b, step.new_node, step.discount, step.vps, step.values, step.logits, step.horizon,
                                      step.a, step.delta, step.vtp, reinterpret_cast<float4 *>(smem), s_sel, step.ts,
                                      heads_on ? s_leaf : nullptr, s_ctr + 2, 3);
            if constexpr (HEADS) __builtin_amdgcn_s_setprio(0);
        } else {
            if constexpr (HEADS) {
                const int hw = wv < 4 ? wv - 1 : (wv == 4 ? 6 : wv - 2);
                if (heads_on) heads_in_prologue<(RES != 0)>(step.sh, b, step.t.A, hw, lane, s_leaf, s_ctr, s_red, step.ts, RES == 2 ? (size_t)rs.ds * rs.B : 0, RES == 2 ? (size_t)rs.ds * rs.BA : 0);
            }
            load_w0();
            prefetch_weights(wv - 1); prefetch_weights(wv + 6);
            for (int i = tid - 64; i < a.nlayers * 128; i += NTHR - 64) {
                const int L = i >> 7, r = i & 127;
                sSS[i] = (r < 64) ? a.layer[L].scale[r] : a.layer[L].shift[r - 64];
            }
        }
        __syncthreads();
        if (step.ts && b == 0 && tid == 0) lz_stamp_store(step.ts + (5), __builtin_readcyclecounter());
        g_slot = s_sel[0];
        g_action = s_sel[1];
        if (wv == 0) load_w0();
    } else {
        if (a.gather_ix) g_slot = a.gather_ix[b];
        if (a.act_table) g_action = a.action[b];
    }
    // the action table's rows of the output tiles this wave finishes (the dynamics convolution adds them): requested with the latent, so that
    // the layers' only vector-memory traffic is the weight stream and no wait of theirs has to drain it
    f32x4 tvv[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int q = min(widx + 4 * f, 5), ntl = q / 3, mt = q - 3 * ntl;
        const int m = min(mt * 16 + (lane & 15), HW - 1), c4 = (2 * np + ntl) * 16 + 4 * (lane >> 4);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        tvv[f] = a.act_table ? *reinterpret_cast<const f32x4 *>(a.act_table + (size_t)g_action * HW * 64 + m * 64 + c4) : z;
    }
    {
        const float *src = a.in + (size_t)b * HW * 64 + (size_t)g_slot * a.slot_stride;
        constexpr int NU = (HW * 16 + NTHR - 1) / NTHR;
        float4 v[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int idx = min(u * NTHR + tid, HW * 16 - 1);
            v[u] = *reinterpret_cast<const float4 *>(src + (size_t)idx * 4);
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int idx = u * NTHR + tid;
            if (idx < HW * 16) {
                *reinterpret_cast<float4 *>(smem + (idx >> 4) * PS + (idx & 15) * 4) = v[u];
                bf16x4 h, m3, l3;
                split3_bf16((f32x4){v[u].x, v[u].y, v[u].z, v[u].w}, h, m3, l3);
                *reinterpret_cast<bf16x4 *>(sB + (idx >> 4) * PB + (idx & 15) * 4) = h;
                *reinterpret_cast<bf16x4 *>(sB + BB + (idx >> 4) * PB + (idx & 15) * 4) = m3;
                *reinterpret_cast<bf16x4 *>(sB + 2 * BB + (idx >> 4) * PB + (idx & 15) * 4) = l3;
            }
        }
        // the all-zero pixel of every buffer (fp32: the head convolutions' padding rows; bf16: the halo of every plane)
        if (tid < 64) *reinterpret_cast<float4 *>(smem + (tid >> 4) * BUF + HW * PS + (tid & 15) * 4) = vzero4();
        {   // 12 (buffer, plane) pairs: everything behind the pixels (the zero line included) in 16-byte pieces
            constexpr int NZ = (int)(LZ_S3_PLANE - (size_t)HW * PB * 2) / 16;
            for (int i = tid - 64; i >= 0 && i < 4 * NPL * NZ; i += NTHR - 64)
                *reinterpret_cast<float4 *>(sB + (i / NZ) * BB + HW * PB + (i % NZ) * 8) = vzero4();
        }
        if (TREE == 0) {
            for (int i = tid; i < a.nlayers * 128; i += NTHR) {
                const int L = i >> 7, r = i & 127;
                sSS[i] = (r < 64) ? a.layer[L].scale[r] : a.layer[L].shift[r - 64];
            }
        }
    }
    // ---- per-lane geometry, the same for every layer: B columns of this lane = pixels 16 mt + (lane & 15); tap (dy, dx) reads pixel
    // m + dy GW + dx when it is inside the image, the zero pixel otherwise (one validity bit per (row tile, tap))
    unsigned long long valid = 0;
    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + (lane & 15), y = m / GW, x = m - y * GW;
        abase[mt] = (m * PB + kh * 32 + (lane >> 4) * 8) * 2;   // bytes
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            if (m < HW && yy >= 0 && yy < GH && xx >= 0 && xx < GW) valid |= 1ull << (mt * 9 + t);
        }
    }
    const int kq4 = (lane >> 4) * 4, zoff = HW * PS;
    __syncthreads();
    if constexpr (TREE != 0) { if (step.ts && b == 0 && tid == 0) lz_stamp_store(step.ts + (6), __builtin_readcyclecounter()); }

    // The MFMAs run TRANSPOSED (weights as the A operand): D[channel][pixel], so a lane ends up with four consecutive channels of ONE pixel --
    // 16 contiguous bytes in every [pixel][channel] array.  Output tile q = 3 ntl + mt of this wave's pair (ntl = 0 | 1); tile q is FINISHED by
    // the wave with widx == q % 4 (two tiles for widx 0 and 1, one for 2 and 3); the other three leave their partial sums in
    // sP[np][q][rank among the others]
    const int nlayers = a.nlayers;
    for (int L = 0; L < nlayers; ++L) {
        const int flags = __builtin_amdgcn_readlane(my_flags, L);
        const int Ln = L + 1 < nlayers ? L + 1 : L;
        const char *sBin = reinterpret_cast<const char *>(sB + (flags & 3) * BB3);
        float *sOut = smem + ((flags >> 2) & 3) * BUF;
        __bf16 *sBout = sB + ((flags >> 2) & 3) * BB3;
        gbl_bf16x8 *wl_cur = as_global_bf16x8(lane64(my_wb, L)), *wl_nxt = as_global_bf16x8(lane64(my_wb, Ln));
        const bool tab = (flags >> 8) & 1, relu = (flags >> 7) & 1;
        const int res = ((flags >> 4) & 7) - 1;
        // operands of the epilogue that do not depend on the products -- folded-BN scale / shift and the residual of the tiles this wave
        // finishes -- are read here: the LDS pipe has room under the products, the epilogue is bound by it (24 registers held for it)
        const float *sRes = smem + max(res, 0) * BUF;
        f32x4 rvv[2], scv[2], shv[2];
        int mpix[2], c4v[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int q = min(widx + 4 * f, 5), ntl = q / 3, mt = q - 3 * ntl;
            mpix[f] = mt * 16 + (lane & 15);
            c4v[f] = (2 * np + ntl) * 16 + 4 * (lane >> 4);
            {
                const f32x4 rl = *reinterpret_cast<const f32x4 *>(sRes + min(mpix[f], HW - 1) * PS + c4v[f]), z = {0.f, 0.f, 0.f, 0.f};
                rvv[f] = res >= 0 ? rl : z;
            }
            scv[f] = *reinterpret_cast<const f32x4 *>(sSS + L * 128 + c4v[f]);
            shv[f] = *reinterpret_cast<const f32x4 *>(sSS + L * 128 + 64 + c4v[f]);
        }
        // pixel fragments of a tap: 3 row tiles of one plane
        auto read_x = [&](int t, int pl, bf16x8 (&af)[NPL][MT]) {
            const int toff = ((t / 3 - 1) * GW + (t % 3 - 1)) * PB * 2;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int offv = abase[mt] + toff;
                const int off = (valid >> (mt * 9 + t)) & 1 ? offv : (ZLINE | (offv & 0xf0));   // the zero line, at this lane's own bank quad
                af[pl][mt] = *reinterpret_cast<const bf16x8 *>(sBin + off + pl * BB * 2);
            }
        };
        f32x4 acc[2][MT];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[n][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // One tap = six of the nine cross products (hi mid lo = planes 0 1 2; the three left out are below 2^-32 of the result), 6 MFMAs each,
        // consecutive MFMAs writing different accumulators.  Neither the pixel fragments nor the ring slot are double-buffered (they would
        // not fit: 256 registers at two waves per SIMD): the order of the products frees x[lo] after the first, w[hi] after the third, x[mid]
        // after the fourth, w[mid] after the fifth, and each is re-requested right there -- the next tap's pixel plane from LDS (two to five
        // products ahead of its use), the ring slot's next occupant from L2 (this layer's tap RT further on, or, behind the slot's last use in
        // this layer, the next layer's tap): w[hi], the first plane the tap after next needs, gets a lead of a tap and a half (with the hi x hi
        // product last it had one tap, 576 cycles for a wave running alone at the end of a layer: less than a loaded L2 round trip)
        auto prod = [&](const bf16x8 (&w)[2][NPL], int wp, const bf16x8 (&x)[NPL][MT], int xp) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[n][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[n][wp], x[xp][mt], acc[n][mt], 0, 0, 0);
        };
        auto taps = [&](auto ntaps_c, int tb) {
            constexpr int NTAP = decltype(ntaps_c)::value;
            bf16x8 x[NPL][MT];
#pragma unroll
            for (int pl = NPL - 1; pl >= 0; --pl) read_x(tb, pl, x);
#pragma unroll
            for (int i = 0; i < NTAP; ++i) {
                bf16x8 (&w)[2][NPL] = wr[i % RT];
                const bool more = i + 1 < NTAP;
                gbl_bf16x8 *wl = (i + RT < NTAP) ? wl_cur : wl_nxt;
                const int wt = (i + RT < NTAP) ? tb + i + RT : tb + i % RT;
#if defined(LZ_DEBUG_KNOBS) && defined(LZ_DEBUG_S3_SKIP)   // timing experiments (results are then wrong; the branches also cost ~60 % per layer): 16 = no weight stream, 4 = no MFMAs, 32 = no pixel-fragment reads
                const bool dW = !(a.debug_flags & 16), dM = !(a.debug_flags & 4), dX = !(a.debug_flags & 32);
#else
                constexpr bool dW = true, dM = true, dX = true;
#endif
                if (dM) prod(w, 0, x, 2);                           // hi  x lo
                __builtin_amdgcn_sched_barrier(0);
                if (more && dX) read_x(tb + i + 1, 2, x);
                if (dM) prod(w, 0, x, 1);                           // hi  x mid
                if (dM) prod(w, 0, x, 0);                           // hi  x hi
                __builtin_amdgcn_sched_barrier(0);
                if (dW) load_w(wl, wt, 0, w);
                if (dM) prod(w, 1, x, 1);                           // mid x mid
                __builtin_amdgcn_sched_barrier(0);
                if (more && dX) read_x(tb + i + 1, 1, x);
                if (dM) prod(w, 1, x, 0);                           // mid x hi
                __builtin_amdgcn_sched_barrier(0);
                if (dW) load_w(wl, wt, 1, w);
                if (dM) prod(w, 2, x, 0);                           // lo  x hi
                __builtin_amdgcn_sched_barrier(0);
                if (dW) load_w(wl, wt, 2, w);
                if (more && dX) read_x(tb + i + 1, 0, x);
            }
        };
        if constexpr (TREE != 0) { if (step.ts && b == 0 && tid == 0 && L == 2) lz_stamp_store(step.ts + (25), __builtin_readcyclecounter()); }
        if (th == 0) taps(std::integral_constant<int, 5>{}, 0);
        else taps(std::integral_constant<int, 4>{}, 5);
        if constexpr (TREE != 0) { if (step.ts && b == 0 && tid == 0 && L == 2) lz_stamp_store(step.ts + (26), __builtin_readcyclecounter()); }
        // ---- the four partial sums of an output tile meet in LDS: every wave leaves the tiles it does not finish
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int fin = q & 3;
            if (fin != widx) {
                const int rank = widx < fin ? widx : widx - 1;
                *reinterpret_cast<f32x4 *>(sP + (((np * 6 + q) * 3 + rank) * 64 + lane) * 4) = acc[q / 3][q % 3];
            }
        }
        __syncthreads();
        if constexpr (TREE != 0) { if (step.ts && b == 0 && tid == 0 && L == 2) lz_stamp_store(step.ts + (27), __builtin_readcyclecounter()); }
        {
            float *go = reinterpret_cast<float *>(lane64(my_gout, L));
            if (go) go += (size_t)b * HW * 64 + (RES == 2 ? (size_t)rs.ds * (size_t)rs.lat_step : (size_t)0);
            // The epilogue is VALU work on the SIMD both waves of a pair share (3 tiles per SIMD): with the wave's role a run-time value every
            // accumulator / partial-sum pick was a chain of v_cndmask (about 120 VALU instructions per tile, 1.4 k cycles per layer on the
            // busiest SIMD) -- so the role becomes a compile-time constant behind a scalar branch, the residual arrives masked, the
            // action-table rows enter by an FMA with 0 | 1 and ReLU is a max with 0 | -inf: about 50.
            const float tabf = tab ? 1.0f : 0.0f, floor_ = relu ? 0.0f : -__builtin_inff();
            auto finish = [&](auto widx_c) {
                constexpr int W = decltype(widx_c)::value;
                constexpr int NFIN = W < 2 ? 2 : 1;          // waves 2 and 3 of a pair finish one tile
                // every LDS read of the epilogue before its first write (the compiler must assume they alias)
                f32x4 oth[NFIN][3];
#pragma unroll
                for (int f = 0; f < NFIN; ++f)
#pragma unroll
                    for (int r = 0; r < 3; ++r) oth[f][r] = *reinterpret_cast<const f32x4 *>(sP + (((np * 6 + W + 4 * f) * 3 + r) * 64 + lane) * 4);
#pragma unroll
                for (int f = 0; f < NFIN; ++f) {
                    const int q = W + 4 * f;
                    // the four partial sums in the fixed order of the waves' indices (kh + 2 th): the finisher's own stands at position W
                    f32x4 p[4];
#pragma unroll
                    for (int w4 = 0; w4 < 4; ++w4) p[w4] = (w4 == W) ? acc[q / 3][q % 3] : oth[f][w4 < W ? w4 : w4 - 1];
                    f32x4 o;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float v = ((p[0][c] + p[1][c]) + p[2][c]) + p[3][c];
                        v = __builtin_fmaf(tabf, tvv[f][c], v);
                        v = v * scv[f][c] + shv[f][c];
                        v += rvv[f][c];
                        o[c] = fmaxf(v, floor_);
                    }
                    bf16x4 oh, om, ol;
                    split3_bf16(o, oh, om, ol);
                    if (mpix[f] < HW) {
                        *reinterpret_cast<f32x4 *>(sOut + mpix[f] * PS + c4v[f]) = o;
                        *reinterpret_cast<bf16x4 *>(sBout + mpix[f] * PB + c4v[f]) = oh;
                        *reinterpret_cast<bf16x4 *>(sBout + BB + mpix[f] * PB + c4v[f]) = om;
                        *reinterpret_cast<bf16x4 *>(sBout + 2 * BB + mpix[f] * PB + c4v[f]) = ol;
                        if (go) { if constexpr (RES == 2) *reinterpret_cast<f32x4 *>(go + mpix[f] * 64 + c4v[f]) = o; else store_wt(go + mpix[f] * 64 + c4v[f], o); }
                    }
                }
            };
            switch (__builtin_amdgcn_readfirstlane(widx)) {
            case 0: finish(std::integral_constant<int, 0>{}); break;
            case 1: finish(std::integral_constant<int, 1>{}); break;
            case 2: finish(std::integral_constant<int, 2>{}); break;
            default: finish(std::integral_constant<int, 3>{}); break;
            }
        }
        if constexpr (TREE != 0) { if (step.ts && b == 0 && tid == 0 && L == 2) lz_stamp_store(step.ts + (29), __builtin_readcyclecounter()); }
        __syncthreads();
        if constexpr (TREE != 0) { if (step.ts && b == 0 && tid == 0 && L < 8) lz_stamp_store(step.ts + (16 + L), __builtin_readcyclecounter()); }
    }
    // 1x1 head convolutions (64 -> 16) + bias + BN + ReLU in fp32, as in k_chain_w
    auto c1_store = [&](const lz_c1_job &jb, int row, int cq, const f32x4 &acc) {
        float4 v;
        v.x = fmaxf((acc[0] + c1b.x) * c1s.x + c1t.x, 0.0f);
        v.y = fmaxf((acc[1] + c1b.y) * c1s.y + c1t.y, 0.0f);
        v.z = fmaxf((acc[2] + c1b.z) * c1s.z + c1t.z, 0.0f);
        v.w = fmaxf((acc[3] + c1b.w) * c1s.w + c1t.w, 0.0f);
        float *po = jb.out + ((size_t)b * HW + row) * jb.out_stride + jb.out_off + cq * 4;
        if constexpr (RES == 2) *reinterpret_cast<f32x4 *>(po) = (f32x4){v.x, v.y, v.z, v.w};   // (the group's LSTM phase reads these rows from this XCD's L2)
        else store_wt(po, (f32x4){v.x, v.y, v.z, v.w});
    };
    auto c1_tile = [&](int job, int i) {
        const float *sIn = smem + a.c1_in[job] * BUF + kq4;
        const int row = i * 16 + (lane & 15);
        const int off = (row < HW) ? row * PS : zoff;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 xf = *reinterpret_cast<const float4 *>(sIn + off + g * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vget(c1w[g], j), vget(xf, j), acc, 0, 0, 0);
        }
        if (row < HW) c1_store(a.c1[job], row, lane >> 4, acc);
    };
    auto c1_rem = [&](int job) {
        const float *sIn = smem + a.c1_in[job] * BUF + kq4;
        const int row = (HW / 16) * 16 + (lane & 3);
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 xf = *reinterpret_cast<const float4 *>(sIn + row * PS + g * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(vget(c1w[g], j), vget(xf, j), acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = xor32_sum(xor16_sum(acc[q]));
        const int cg = (lane >> 2) & 3;
        if ((lane >> 4) == cg) c1_store(a.c1[job], row, cg, acc);
    };
    if constexpr (C1SPLIT) {
        if (wv < a.nc1) { c1_tile(c1j, 0); c1_rem(c1j); }
        else if (wv < 2 * a.nc1) c1_tile(c1j, 1);
    } else if (wv < a.nc1) {
#pragma unroll
        for (int i = 0; i < MT; ++i) c1_tile(wv, i);
    }
    if constexpr (TREE != 0) { if (step.ts && b == 0 && tid == 0) lz_stamp_store(step.ts + (24), __builtin_readcyclecounter()); }
}

template <int GW, int GH, int TREE = 0, bool HEADS = false>
__global__ __launch_bounds__(512) void k_chain_s3(lz_chain_args a, typename step_arg<TREE>::type step)
{
    lz_stamp_begin(a.stamp);
    chain_s3_body<GW, GH, TREE, HEADS, 0>(a, step, (int)blockIdx.x, (int)gridDim.x, lz_res_sim{});
    lz_stamp_end(a.stamp);
}

// ------------------------------------------------------------------------------------------------
// FAST MODE: 3x3 convolution of the representation tower (48^2, 24^2, 12^2 grids; stride 1 | 2) on v_mfma_f32_16x16x32_bf16.
// One tile = TR output rows x the full width of one image = 96 output pixels (six 16-pixel MFMA column tiles); the input halo of the tile
// -- (TR - 1) STRIDE + 3 rows x (Wout - 1) STRIDE + 3 columns x CIN -- is copied from the bf16 NHWC tensor and laid out
// [row][column][CIN + pad] in LDS (pixel pitch 6 | 10 | 5 bank quads: conflict-free ds_read_b128 for stride 1 | 2).  The MFMAs run
// transposed (weights = A operand, pixels = B operand: D[channel][pixel]), so a lane ends with four consecutive output channels of one
// pixel and the epilogue (BatchNorm, residual, ReLU) stores them together.  Workgroups are PERSISTENT: wave (nt, mg) keeps all
// 9 CIN / 32 weight fragments of its 16-channel tile nt in registers and walks the tiles blockIdx.x, + gridDim.x, ...; the NEXT tile's halo is
// requested into registers before this tile's MFMAs and lands in the other of two LDS buffers (one barrier per tile).  fp32 accumulation; the
// tower's activations are bf16 NHWC tensors in HBM in this mode (lz_conv_args::act_bf16: input, residual, output).
// ------------------------------------------------------------------------------------------------
template <int CIN, int COUT, int STRIDE>
__global__ __launch_bounds__(256) void k_conv_bf(lz_conv_args a, int ntiles, int TR)
{
    constexpr int NT = COUT / 16, MG = 4 / NT;          // waves = NT channel tiles x MG pixel groups
    constexpr int MTW = 6 / MG;                         // 16-pixel tiles per wave (96 pixels per workgroup tile)
    constexpr int KC = CIN / 32, KS = 9 * KC;           // k steps of 32: (tap, 32-channel block)
    constexpr int PBq = STRIDE == 2 ? 5 : (CIN == 32 ? 6 : 10), PB = PBq * 8;   // pixel pitch in bf16
    constexpr int C8 = CIN / 8, NLD = STRIDE == 2 ? 7 : 5;   // 16-byte pieces per pixel; pieces per thread and halo (launch_conv_bf checks the bound)
    static_assert((CIN == 32 || CIN == 64) && (COUT == 32 || COUT == 64), "shapes of the tower");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nt = wv % NT, mg = wv / NT;
    const int Wout = a.Wout, Hout = a.Hout, Win = a.Win, Hin = a.Hin;
    const int HR = (TR - 1) * STRIDE + 3, HC = (Wout - 1) * STRIDE + 3;
    const int bands = (Hout + TR - 1) / TR;
    const int n8 = HR * HC * C8;
    __bf16 *sH0 = reinterpret_cast<__bf16 *>(smem);     // two halo buffers [HR][HC][PB]
    const int hbuf = ((HR * HC * PB + 7) & ~7);
    const __bf16 *in = reinterpret_cast<const __bf16 *>(a.in), *res = reinterpret_cast<const __bf16 *>(a.residual);
    __bf16 *out = reinterpret_cast<__bf16 *>(a.out);
    // ---- this wave's weights: [nt][ks][64 lanes][8 bf16]
    bf16x8 wq[KS];
    {
        const bf16x8 *wp = reinterpret_cast<const bf16x8 *>(a.wb) + (size_t)nt * KS * 64 + lane;
#pragma unroll
        for (int k = 0; k < KS; ++k) wq[k] = wp[(size_t)k * 64];
    }
    const int co4 = nt * 16 + 4 * (lane >> 4);
    const f32x4 sc = *reinterpret_cast<const f32x4 *>(a.scale + co4), sh = *reinterpret_cast<const f32x4 *>(a.shift + co4);
    // pixel geometry of this lane's B columns (the same for every tile): pixel p = 16 (mg MTW + i) + (lane & 15) of the tile
    int pbase[MTW], prow[MTW], pcol[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int p = 16 * (mg * MTW + i) + (lane & 15);
        prow[i] = p / Wout; pcol[i] = p - prow[i] * Wout;
        pbase[i] = ((prow[i] * STRIDE) * HC + pcol[i] * STRIDE) * PB + (lane >> 4) * 8;   // halo position of tap (0, 0), this lane's k group
    }
    // halo geometry of this thread's 16-byte pieces (the same for every tile): piece u = (pixel, 8-channel block)
    int hsrc[NLD], hdst[NLD], hrow[NLD];
    bool hcol_ok[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int idx = min(u * 256 + tid, n8 - 1);
        const int pix = idx / C8, c8 = idx - pix * C8, hr = pix / HC, hc = pix - hr * HC, ix = hc - 1;
        hrow[u] = hr;
        hcol_ok[u] = (ix >= 0) & (ix < Win);
        hsrc[u] = min(max(ix, 0), Win - 1) * CIN + c8 * 8;
        hdst[u] = (u * 256 + tid < n8) ? pix * PB + c8 * 8 : -1;
    }
    bf16x8 pv[NLD];
    auto prefetch = [&](int tile) {   // the whole halo of a tile in flight at once (clamped addresses, zero outside the image by select)
        const int img = tile / bands, band = tile - img * bands, iy0 = band * TR * STRIDE - 1;
        const __bf16 *src = in + (size_t)img * Hin * Win * CIN;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int iy = iy0 + hrow[u];
            const bf16x8 t = *reinterpret_cast<const bf16x8 *>(src + (size_t)min(max(iy, 0), Hin - 1) * Win * CIN + hsrc[u]);
            const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            pv[u] = ((iy >= 0) & (iy < Hin) & hcol_ok[u]) ? t : z;
        }
    };
    int tile = blockIdx.x, buf = 0;
    if (tile < ntiles) prefetch(tile);
    for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        const int img = tile / bands, band = tile - img * bands;
        const int oy0 = band * TR;
        __bf16 *sH = sH0 + buf * hbuf;
        // ---- the prefetched halo into this tile's LDS buffer (double-buffered: one barrier per tile)
#pragma unroll
        for (int u = 0; u < NLD; ++u)
            if (hdst[u] >= 0) *reinterpret_cast<bf16x8 *>(sH + hdst[u]) = pv[u];
        // ---- residual rows of this lane's outputs (consumed in the epilogue)
        bf16x4 rv[MTW];
        bool pok[MTW];
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            pok[i] = oy0 + prow[i] < Hout;
            const size_t o = (((size_t)img * Hout + min(oy0 + prow[i], Hout - 1)) * Wout + pcol[i]) * COUT + co4;
            const bf16x4 z = {0, 0, 0, 0};
            rv[i] = res ? *reinterpret_cast<const bf16x4 *>(res + o) : z;
        }
        __syncthreads();
        // ---- the next tile's halo travels while this one is multiplied
        if (tile + (int)gridDim.x < ntiles) prefetch(tile + gridDim.x);
        // ---- products: tap by tap, the MTW pixel fragments of a k step are requested together
        f32x4 acc[MTW];
#pragma unroll
        for (int i = 0; i < MTW; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int toff = ((t / 3) * HC + (t % 3)) * PB;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                bf16x8 bf[MTW];
#pragma unroll
                for (int i = 0; i < MTW; ++i) bf[i] = *reinterpret_cast<const bf16x8 *>(sH + pbase[i] + toff + kc * 32);
#pragma unroll
                for (int i = 0; i < MTW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[t * KC + kc], bf[i], acc[i], 0, 0, 0);
            }
        }
        // ---- epilogue: BatchNorm, residual, ReLU; four consecutive channels of one pixel per lane
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            bf16x4 ob;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = acc[i][q] * sc[q] + sh[q];
                v += (float)rv[i][q];
                ob[q] = (__bf16)(a.relu ? fmaxf(v, 0.0f) : v);
            }
            if (pok[i]) *reinterpret_cast<bf16x4 *>(out + (((size_t)img * Hout + oy0 + prow[i]) * Wout + pcol[i]) * COUT + co4) = ob;
        }
    }
}

template <int CIN, int COUT, int STRIDE>
static void launch_conv_bf(const lz_conv_args &a, hipStream_t s)
{
    const int TR = 96 / a.Wout;
    const int HR = (TR - 1) * STRIDE + 3, HC = (a.Wout - 1) * STRIDE + 3;
    constexpr int PB = (STRIDE == 2 ? 5 : (CIN == 32 ? 6 : 10)) * 8;
    const size_t lds = (size_t)2 * (((size_t)HR * HC * PB + 7) & ~(size_t)7) * 2;
    const int ntiles = a.B * ((a.Hout + TR - 1) / TR);
    const int grid = ntiles < 512 ? ntiles : 512;   // persistent: two workgroups per CU
    hipLaunchKernelGGL((k_conv_bf<CIN, COUT, STRIDE>), dim3(grid), dim3(256), lds, s, a, ntiles, TR);
}

// ------------------------------------------------------------------------------------------------
// PARITY MODE (fp32 accuracy), round 5: the tower's 3x3 convolutions as SPLIT-bf16 products on v_mfma_f32_16x16x32_bf16.
// An fp32 number is EXACTLY the sum of three bf16 numbers (hi = rne(x), mid = rne(x - hi), lo = rne(x - hi - mid): 3 x 8 significant
// bits).  With both operands split, x w = sum of nine bf16 x bf16 products -- each exact in the fp32 accumulator -- of which the three
// smallest (mid lo, lo mid, lo lo: below 2^-24 of the product) are dropped: six MFMAs per k-step of 32, fp32 accumulation.  The error of
// a dot product is that of an fp32 FMA chain (measured on a 576-term product against binary64: 7e-7 relative, torch's own fp32 GEMM
// 1.3e-6); the tower's output stays inside north_star's 1e-5 (1 + |x|) of the reference modules (tests/test_nn_golden_gpu.py,
// test_nn_gpu.py) -- this is NOT the fast mode (one bf16 product, statistical parity only).  Why: the bf16 pipe runs 16x the fp32-matrix
// rate, so six products cost 0.375 of the direct fp32 form and 0.84 of the Winograd F(2x2, 3x3) form these layers used (k_conv_wino), without
// its transforms (VALU + LDS work that was half of those kernels' time): the tower is compute-bound (weights are reused over thousands of
// pixels), unlike the recurrent chain and the LSTM, which are bound by their weight stream -- there three planes of bf16 are MORE bytes
// than fp32 Winograd weights and the split would lose.
// Structure = k_conv_bf: persistent workgroups, tile = TR output rows x the image width = 96 output pixels, the halo of a tile staged
// in LDS ([3 planes][row][column][CIN + pad] bf16, conflict-free pixel pitch), transposed MFMAs (weights = A operand), BatchNorm /
// residual / ReLU on four consecutive channels of a pixel per lane.  Differences: activations are the parity mode's fp32 NHWC tensors
// (split while they are staged: the NEXT tile's raw halo is requested into registers before this tile's products), and the weights --
// three planes, [nt][k step][plane][64 lanes][8 bf16] -- stream from L2 one k-step ahead (216 registers would be needed to keep a
// 64-channel layer's resident).
// ------------------------------------------------------------------------------------------------
// DUAL (round 6; the downsample block's conv1 and its shortcut conv3, common.py:314: both 3x3 stride 2 on the SAME input): one launch of
// 512-thread workgroups computes both -- waves 0-3 the first convolution (a), waves 4-7 the second (d: its own weights, BatchNorm, ReLU flag
// and output) on the halo that is staged ONCE.  The stride-2 instance is one workgroup per CU (106 KB of halo planes): per tile it spent
// 19 k cycles for 5.2 k cycles of matrix work, most of them staging; the second convolution rides on the same staging.
struct lz_conv_second { const void *w3; const float *scale, *shift; float *out; int relu; };
template <int CIN, int COUT, int STRIDE, bool DUAL = false>
__global__ __launch_bounds__(DUAL ? 512 : 256) __attribute__((amdgpu_waves_per_eu(DUAL ? 2 : (STRIDE == 2 ? 1 : 2)))) void k_conv_s3(lz_conv_args a, int ntiles, int TR, lz_conv_second d)
{
    // Wave roles.  A wave owns TWO 16-channel output tiles for three 16-pixel tiles: every pixel fragment read from LDS feeds 4 MFMAs
    // (with one channel tile per wave it fed 2, and the LDS read pipe -- 324 KB per 96-pixel tile, 2.5 k cycles at 128 B/clk -- was as
    // busy as the matrix pipe).  64 output channels: waves = 2 channel pairs x 2 pixel groups.  32 output channels (one pair): waves =
    // 2 pixel groups x 2 halves of the taps (0-4 | 5-8), the two partial sums of an output tile meet in LDS, always in the order
    // taps 0-4 + taps 5-8.
    constexpr int NP = COUT / 32;
    constexpr bool TSPLIT = NP == 1;
    constexpr int KC = CIN / 32, KS = 9 * KC;           // k steps of 32: (tap, 32-channel block)
    constexpr int PBq = STRIDE == 2 ? 5 : (CIN == 32 ? 6 : 10), PB = PBq * 8;   // pixel pitch in bf16 (k_conv_bf's: conflict-free ds_read_b128)
    constexpr int NTHR = DUAL ? 512 : 256;
    constexpr int C4 = CIN / 4, NLD = STRIDE == 2 ? (DUAL ? 7 : 14) : (CIN == 32 ? 7 : 10);  // 16-byte fp32 pieces per pixel; pieces per thread and halo (the launcher checks the bound)
    static_assert((CIN == 32 || CIN == 64) && (COUT == 32 || COUT == 64), "shapes of the tower");
    static_assert(!DUAL || (COUT == 64 && STRIDE == 2), "the dual form is the downsample block's");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = (tid >> 6) & 3, second = DUAL ? (tid >> 8) : 0;   // second: this wave computes the second convolution
    const int np = TSPLIT ? 0 : (wv & 1), mg = TSPLIT ? (wv & 1) : (wv >> 1);
    // (two workgroups share a CU and wave i of both sits on SIMD i: the second one swaps the tap halves so that every SIMD gets 5 + 4 taps)
    const int th = TSPLIT ? ((wv >> 1) ^ ((blockIdx.x >> 8) & 1)) : 0;
    const int ks_begin = TSPLIT ? (th ? 5 * KC : 0) : 0, ks_end = TSPLIT ? (th ? KS : 5 * KC) : KS;
    const int Wout = a.Wout, Hout = a.Hout, Win = a.Win, Hin = a.Hin;
    const int HR = (TR - 1) * STRIDE + 3, HC = (Wout - 1) * STRIDE + 3;
    const int bands = (Hout + TR - 1) / TR;
    const int n4 = HR * HC * C4;
    const int hplane = ((HR * HC * PB + 7) & ~7);       // bf16 per plane
    __bf16 *sH = reinterpret_cast<__bf16 *>(smem);      // [3 planes][HR][HC][PB]
    float *sX = reinterpret_cast<float *>(sH + 3 * hplane);   // TSPLIT: [pixel group][sending half][3 tiles][64 lanes][4] partial sums
    const float *in = a.in, *res = second ? nullptr : a.residual;
    float *out = second ? d.out : a.out;
    const bool relu_on = (second ? d.relu : a.relu) != 0;
    const bf16x8 *wp = reinterpret_cast<const bf16x8 *>(second ? d.w3 : a.w3) + (size_t)(2 * np) * KS * 3 * 64 + lane;   // [nt][ks][plane][64 lanes]; the pair's second tile: + KS * 3 * 64
    // folded-BatchNorm scale | shift wait in LDS for the epilogue (16 registers across the k loop otherwise: the 64-channel instance spilled)
    float *sSS = sX + (TSPLIT ? 4 * 3 * 256 : 0) + second * 2 * COUT;   // [2][COUT] (DUAL: one block per convolution)
    {
        const int t2 = tid & 255;
        const float *scp = second ? d.scale : a.scale, *shp = second ? d.shift : a.shift;
        if (t2 < 2 * COUT) sSS[t2] = t2 < COUT ? scp[t2] : shp[t2 - COUT];
    }
    const int co4b = (2 * np) * 16 + 4 * (lane >> 4);          // first channel of this lane in the pair's first tile (second: + 16)
    int pbase[3], prow[3], pcol[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int p = 16 * (mg * 3 + i) + (lane & 15);
        prow[i] = p / Wout; pcol[i] = p - prow[i] * Wout;
        pbase[i] = ((prow[i] * STRIDE) * HC + pcol[i] * STRIDE) * PB + (lane >> 4) * 8;   // halo position of tap (0, 0), this lane's k group
    }
    // per piece two words (registers are what bounds this kernel's occupancy): source offset inside a row, and
    // (LDS offset + 1: 0 = no piece) | halo row << 20 | column inside the image << 30
    int hsrc[NLD], hpk[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int idx = min(u * NTHR + tid, n4 - 1);
        const int pix = idx / C4, c4 = idx - pix * C4, hr = pix / HC, hc = pix - hr * HC, ix = hc - 1;
        hsrc[u] = min(max(ix, 0), Win - 1) * CIN + c4 * 4;
        hpk[u] = ((u * NTHR + tid < n4) ? pix * PB + c4 * 4 + 1 : 0) | (hr << 20) | (((ix >= 0) & (ix < Win)) ? (1 << 30) : 0);
    }
    f32x4 pv[NLD];
    auto prefetch = [&](int tile) {   // the whole fp32 halo of a tile in flight at once (clamped addresses, zero outside the image by select)
        const int img = tile / bands, band = tile - img * bands, iy0 = band * TR * STRIDE - 1;
        const float *src = in + (size_t)img * Hin * Win * CIN;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int iy = iy0 + ((hpk[u] >> 20) & 0x3ff);
            const f32x4 t = *reinterpret_cast<const f32x4 *>(src + (size_t)min(max(iy, 0), Hin - 1) * Win * CIN + hsrc[u]);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            pv[u] = ((iy >= 0) & (iy < Hin) & ((hpk[u] >> 30) & 1)) ? t : z;
        }
    };
    auto koff = [&](int ks) {         // halo offset (bf16) of k step ks = (tap, 32-channel block)
        const int t = ks / KC, kc = ks - t * KC, ty = t / 3;
        return (ty * HC + (t - 3 * ty)) * PB + kc * 32;
    };
    // Tile order: the workgroups of an XCD (block id % 8) walk ONE contiguous eighth of the tiles together, so the bands that share halo
    // rows (a tile re-reads (HR - TR STRIDE) / HR of its neighbours' rows: half of them at TR = 2) are in flight on the same L2 at the same time;
    // dealt round-robin, neighbouring bands sat on different XCDs and the overlap was fetched again from the memory side
    const bool xcd_order = (gridDim.x & 7) == 0 && ntiles >= 64;
    const int tper = xcd_order ? (ntiles + 7) >> 3 : ntiles, tstride = xcd_order ? (int)gridDim.x >> 3 : (int)gridDim.x;
    const int tbase = xcd_order ? (blockIdx.x & 7) * tper : 0, tend = min(tbase + tper, ntiles);
    int tile = tbase + (xcd_order ? (int)blockIdx.x >> 3 : (int)blockIdx.x);
    if (tile < tend) prefetch(tile);
    for (; tile < tend; tile += tstride) {
        const int img = tile / bands, band = tile - img * bands;
        const int oy0 = band * TR;
        // ---- the prefetched halo, split into its three bf16 planes, into LDS (one buffer: the barrier below the products frees it)
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            bf16x4 h, m, l;
            split3_bf16(pv[u], h, m, l);
            const int hd = (hpk[u] & 0xfffff) - 1;
            if (hd >= 0) {
                *reinterpret_cast<bf16x4 *>(sH + hd) = h;
                *reinterpret_cast<bf16x4 *>(sH + hplane + hd) = m;
                *reinterpret_cast<bf16x4 *>(sH + 2 * hplane + hd) = l;
            }
        }
        // the first k-step's weights (two channel tiles x three planes)
        bf16x8 wc[2][3], wn[2][3];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wc[n][pl] = wp[((size_t)n * KS + ks_begin) * 3 * 64 + (size_t)pl * 64];
        __syncthreads();
        // ---- the next tile's halo travels while this one is multiplied
        if (tile + tstride < tend) prefetch(tile + tstride);
        // The residual rows of the first channel tile this wave finishes are requested here where registers allow (32 output channels,
        // stride 2): they do not depend on the products, and requested in the epilogue their round trip (another kernel wrote them) stands
        // between the products and the store of every tile with only the co-resident workgroup to hide it (32 -> 32 at 48 x 48: 74.5 -> 71.9 us).
        // The 64 -> 64 stride-1 instance has no 12 registers left across the k loop (it would spill): both of its tiles' rows are requested
        // at the top of the epilogue.
        auto res_request = [&](int c4o, f32x4 (&rv)[3]) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const size_t o = (((size_t)img * Hout + min(oy0 + prow[i], Hout - 1)) * Wout + pcol[i]) * COUT + c4o;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                rv[i] = res ? *reinterpret_cast<const f32x4 *>(res + o) : z;
            }
        };
        constexpr bool EARLY = TSPLIT || STRIDE == 2;
        f32x4 rv0[3];
        if constexpr (EARLY) res_request(co4b + 16 * (TSPLIT ? th : 0), rv0);
        f32x4 acc[2][3];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[n][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // pixel fragments x[plane][tile]: single-buffered -- the order of the six products frees the lo plane after the first, the mid plane
        // after the fourth, and the next k-step's are requested right there (k_chain_s3's schedule); the weights stay one k-step ahead
        bf16x8 x[3][3];
        auto read_x = [&](int off, int pl) {
#pragma unroll
            for (int i = 0; i < 3; ++i) x[pl][i] = *reinterpret_cast<const bf16x8 *>(sH + pl * hplane + pbase[i] + off);
        };
        auto prod = [&](int wp_, int xp) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int i = 0; i < 3; ++i) acc[n][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[n][wp_], x[xp][i], acc[n][i], 0, 0, 0);
        };
        {
            const int o0 = koff(ks_begin);
            read_x(o0, 2); read_x(o0, 1); read_x(o0, 0);
        }
        // The k loop is NOT unrolled (registers): per k-step 6 weight fragments (one step ahead), 9 LDS reads, 36 MFMAs
#pragma unroll 1
        for (int ks = ks_begin; ks < ks_end; ++ks) {
            const int ksn = min(ks + 1, ks_end - 1);
            const int on = koff(ksn);
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wn[n][pl] = wp[((size_t)n * KS + ksn) * 3 * 64 + (size_t)pl * 64];
            // six of the nine cross products (hi mid lo = planes 0 1 2); consecutive MFMAs write different accumulators
            prod(0, 2);                                   // hi  x lo
            __builtin_amdgcn_sched_barrier(0);
            read_x(on, 2);
            prod(1, 1);                                   // mid x mid
            prod(2, 0);                                   // lo  x hi
            prod(0, 1);                                   // hi  x mid
            __builtin_amdgcn_sched_barrier(0);
            read_x(on, 1);
            prod(1, 0);                                   // mid x hi
            prod(0, 0);                                   // hi  x hi
            __builtin_amdgcn_sched_barrier(0);
            read_x(on, 0);
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wc[n][pl] = wn[n][pl];
        }
        // ---- 32 output channels: the wave of tap half h finishes channel tile h; its sums for the other tile go to its partner
        if constexpr (TSPLIT) {
#pragma unroll
            for (int i = 0; i < 3; ++i) *reinterpret_cast<f32x4 *>(sX + (((mg * 2 + th) * 3 + i) * 64 + lane) * 4) = th ? acc[0][i] : acc[1][i];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const f32x4 o = *reinterpret_cast<const f32x4 *>(sX + (((mg * 2 + (1 - th)) * 3 + i) * 64 + lane) * 4);
                const f32x4 mine = th ? acc[1][i] : acc[0][i];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[0][i][q] = th ? o[q] + mine[q] : mine[q] + o[q];   // taps 0-4 first
            }
        }
        // ---- epilogue: BatchNorm, residual, ReLU; four consecutive channels of one pixel per lane
        constexpr int NF = TSPLIT ? 1 : 2;
        f32x4 rv1[3];
        if constexpr (!EARLY) res_request(co4b + 16 * (TSPLIT ? th : 0), rv0);
        if constexpr (NF == 2) res_request(co4b + 16, rv1);
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            const int c4o = co4b + 16 * (TSPLIT ? th : n);
            const f32x4 scn = *reinterpret_cast<const f32x4 *>(sSS + c4o), shn = *reinterpret_cast<const f32x4 *>(sSS + COUT + c4o);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f32x4 ov;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = acc[n][i][q] * scn[q] + shn[q];
                    v += n == 0 ? rv0[i][q] : rv1[i][q];
                    ov[q] = relu_on ? fmaxf(v, 0.0f) : v;
                }
                if (oy0 + prow[i] < Hout) *reinterpret_cast<f32x4 *>(out + (((size_t)img * Hout + oy0 + prow[i]) * Wout + pcol[i]) * COUT + c4o) = ov;
            }
        }
        __syncthreads();   // every wave is done with this tile's halo (and the exchange area): the next one may be written
    }
}

template <int CIN, int COUT, int STRIDE>
static bool launch_conv_s3(const lz_conv_args &a, hipStream_t s)
{
    if (96 % a.Wout != 0) return false;
    const int TR = 96 / a.Wout;
    const int HR = (TR - 1) * STRIDE + 3, HC = (a.Wout - 1) * STRIDE + 3;
    constexpr int PB = (STRIDE == 2 ? 5 : (CIN == 32 ? 6 : 10)) * 8;
    constexpr int NLD = STRIDE == 2 ? 14 : (CIN == 32 ? 7 : 10);
    if (HR * HC * (CIN / 4) > NLD * 256) return false;          // the halo must fit the kernel's per-thread piece count
    const size_t lds = (size_t)3 * (((size_t)HR * HC * PB + 7) & ~(size_t)7) * 2 + (COUT == 32 ? 4 * 3 * 256 * 4 : 0) + 2 * COUT * 4;   // halo planes (+ the tap halves' exchange) + scale | shift
    if (lds > 150 * 1024) return false;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void *)k_conv_s3<CIN, COUT, STRIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr_set = true; }
    const int ntiles = a.B * ((a.Hout + TR - 1) / TR);
    const int per_cu = lds > 76 * 1024 ? 1 : 2;
    const int grid = ntiles < 256 * per_cu ? ntiles : 256 * per_cu;   // persistent
    hipLaunchKernelGGL((k_conv_s3<CIN, COUT, STRIDE>), dim3(grid), dim3(256), lds, s, a, ntiles, TR, lz_conv_second{});
    return true;
}


// ------------------------------------------------------------------------------------------------
// The same chain for narrow networks (num_channels = 32 | 16: the reference's gomoku / tictactoe configs,
// zoo/board_games/gomoku/config/gomoku_muzero_bot_mode_config.py:41-42, tictactoe/...:33-34).  One workgroup per root, activations in
// LDS across layers; C / 16 output-channel tiles, so the four waves split as (N-tile, M-group): with 32 channels two waves share
// the pixel tiles of each channel half, with 16 channels all four split the pixels.  K = 9 taps x C / 16 steps (18 | 9): the
// whole layer's weight fragments sit in registers and the next layer's are requested before this layer's first MFMA.  Small
// boards, small batches: written for correctness and a short launch, not tuned like k_chain; the tree step keeps its own launch.
// ------------------------------------------------------------------------------------------------
template <int GW, int GH, int C>
__global__ __launch_bounds__(256) void k_chain_small(lz_chain_args a)
{
    constexpr int PS = C + 4, HW = GW * GH, MT = (HW + 15) / 16, BUF = (HW + 1) * PS;
    constexpr int NT = C / 16, MG = 4 / NT, G = C / 16, STEPS = 9 * G, MTW = (MT + MG - 1) / MG, C4 = C / 4;
    static_assert(C == 16 || C == 32, "narrow chain: 16 or 32 channels");
    extern __shared__ __attribute__((aligned(16))) float smem[];  // 4 activation buffers, the action-table slice, scale / shift
    float *sTab = smem + 4 * BUF;
    float *sSS = sTab + HW * PS;   // [LZ_CHAIN_MAX_LAYERS][2][C]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, b = blockIdx.x;
    const int nt = wv % NT, mg = wv / NT;
    f32x4 wcur[STEPS], wnxt[STEPS];
    {
        const f32x4 *w0 = reinterpret_cast<const f32x4 *>(a.layer[0].wf) + (size_t)nt * STEPS * 64 + lane;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) wcur[s] = w0[s * 64];
    }
    const int g_slot = a.gather_ix ? a.gather_ix[b] : 0, g_action = a.act_table ? a.action[b] : 0;
    {
        const float *src = a.in + (size_t)b * HW * C + (size_t)g_slot * a.slot_stride;
        for (int idx = tid; idx < HW * C4; idx += 256)
            *reinterpret_cast<float4 *>(smem + (idx / C4) * PS + (idx % C4) * 4) = *reinterpret_cast<const float4 *>(src + (size_t)idx * 4);
        if (tid < 4 * C4) *reinterpret_cast<float4 *>(smem + (tid / C4) * BUF + HW * PS + (tid % C4) * 4) = vzero4();  // the all-zero pixel of every buffer
        if (a.act_table) {
            const float *tsrc = a.act_table + (size_t)g_action * HW * C;
            for (int idx = tid; idx < HW * C4; idx += 256)
                *reinterpret_cast<float4 *>(sTab + (idx / C4) * PS + (idx % C4) * 4) = *reinterpret_cast<const float4 *>(tsrc + (size_t)idx * 4);
        }
        for (int i = tid; i < a.nlayers * 2 * C; i += 256) {
            const int L = i / (2 * C), r = i % (2 * C);
            sSS[i] = (r < C) ? a.layer[L].scale[r] : a.layer[L].shift[r - C];
        }
    }
    // this wave's M-tiles: mg, mg + MG, ...; geometry of this lane's row in each
    const int zoff = HW * PS;
    int base[MTW], mask[MTW];
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
        const int row = (mg + j * MG) * 16 + (lane & 15);
        const int p = min(row, HW - 1), y = p / GW, x = p - y * GW;
        int mk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
            mk |= ((iy >= 0) & (iy < GH) & (ix >= 0) & (ix < GW) & (row < HW)) << t;
        }
        int bs = p * PS;
        asm volatile("" : "+v"(mk), "+v"(bs));
        base[j] = bs;
        mask[j] = mk;
    }
    const int kq4 = (lane >> 4) * 4;
    __syncthreads();
    for (int L = 0; L < a.nlayers; ++L) {
        const lz_chain_layer &ly = a.layer[L];
        const float *sIn = smem + ly.in * BUF + kq4;
        float *sOut = smem + ly.out * BUF;
        {   // the next layer's fragments travel while this layer computes
            const f32x4 *wn = reinterpret_cast<const f32x4 *>(a.layer[min(L + 1, a.nlayers - 1)].wf) + (size_t)nt * STEPS * 64 + lane;
#pragma unroll
            for (int s = 0; s < STEPS; ++s) wnxt[s] = wn[s * 64];
        }
        f32x4 acc[MTW];
#pragma unroll
        for (int j = 0; j < MTW; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int t = s / G, g = s % G;
            const int toff = ((t / 3 - 1) * GW + (t % 3 - 1)) * PS;
#pragma unroll
            for (int j = 0; j < MTW; ++j) {
                const int bit = (mask[j] >> t) & 1;
                const int off = zoff + bit * (base[j] + toff - zoff);
                const float4 af = *reinterpret_cast<const float4 *>(sIn + off + g * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(vget(af, q), wcur[s][q], acc[j], 0, 0, 0);
            }
        }
        const int col = nt * 16 + (lane & 15);
        const float sc = sSS[L * 2 * C + col], sh = sSS[L * 2 * C + C + col];
        const float *sRes = smem + max(ly.res, 0) * BUF;
#pragma unroll
        for (int j = 0; j < MTW; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = (mg + j * MG) * 16 + 4 * (lane >> 4) + q;
                if (row < HW) {
                    float v = acc[j][q];
                    if (ly.act) v += sTab[row * PS + col];
                    v = v * sc + sh;
                    if (ly.res >= 0) v += sRes[row * PS + col];
                    if (ly.relu) v = fmaxf(v, 0.0f);
                    sOut[row * PS + col] = v;
                    if (ly.gout) ly.gout[((size_t)b * HW + row) * C + col] = v;
                }
            }
#pragma unroll
        for (int s = 0; s < STEPS; ++s) wcur[s] = wnxt[s];
        __syncthreads();
    }
    // 1x1 head convolutions (C -> 16) + bias + BN + ReLU: wave j runs job j over every pixel tile
    if (wv < a.nc1) {
        const lz_c1_job &jb = a.c1[wv];
        const float *sIn = smem + a.c1_in[wv] * BUF + kq4;
        const int col = lane & 15;
        const float bi = jb.bias[col], sc = jb.scale[col], sh = jb.shift[col];
        float4 cw[G];
#pragma unroll
        for (int g = 0; g < G; ++g) cw[g] = *reinterpret_cast<const float4 *>(jb.w + (size_t)col * C + g * 16 + kq4);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const int r16 = i * 16 + (lane & 15);
            const int off = (r16 < HW) ? r16 * PS : zoff;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float4 af = *reinterpret_cast<const float4 *>(sIn + off + g * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vget(af, q), vget(cw[g], q), acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = i * 16 + 4 * (lane >> 4) + q;
                if (row < HW) jb.out[((size_t)b * HW + row) * jb.out_stride + jb.out_off + col] = fmaxf((acc[q] + bi) * sc + sh, 0.0f);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LSTM step.  gates[B][4H] = [x | h] . Wcat^T ; tile = 64 rows x 32 gate columns (= 8 hidden units),
// K streamed through LDS in 64-wide chunks (double buffered), 4 waves split each chunk's K.
// grid = (ceil(B/64), 4H/32), block = 256.
// ------------------------------------------------------------------------------------------------
// LSTM cell non-linearities on the hardware exp / rcp (v_exp_f32, v_rcp_f32: ~1 ulp each): the cell epilogue sits on every
// workgroup's critical path and the libm-grade expf / tanhf / IEEE division cost ~400 instructions per (row, unit) pair.
// |error| <= ~3e-7 absolute on values in [-1, 1] (tests compare h, c at 2e-5).  Saturates correctly: exp -> inf => rcp -> 0.
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

template <int NCHUNK, int MROWS>
__global__ __launch_bounds__(256) void k_lstm(lz_lstm_args a)
{
    constexpr int KC = 64, PS = KC + 4, D = 4;  // D chunks of global loads in flight (register ring)
    constexpr int NA = MROWS / 16;              // float4 per thread per A chunk (= M-tiles per workgroup)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    auto sA = [&](int buf) -> float * { return smem + buf * MROWS * PS; };
    auto sB = [&](int buf) -> float * { return smem + 2 * MROWS * PS + buf * 32 * PS; };
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // blockIdx.x walks the gate columns: consecutive workgroups land on different XCDs (block b -> XCD b % 8), so
    // each XCD's L2 fetches only 1/8 of the 8.9 MB weight matrix instead of all of it (71 MB -> 18 MB of fabric reads)
    const int r0 = blockIdx.y * MROWS, n0 = blockIdx.x * 32;
    const int K = a.KX + a.H;
    constexpr int nchunk = NCHUNK;  // K / 64, compile-time: the chunk loop is straight-line code so that the
                                    // compiler's vmcnt bookkeeping keeps D chunks of loads in flight
    const size_t slot = (size_t)a.B * a.H;

    // per-thread source rows (fixed across chunks): thread (row16 = tid >> 4, c4 = tid & 15) loads rows
    // row16 + 16 i of the A chunk (i < NA) and of the B chunk (i < 2)
    const int row16 = tid >> 4, c4 = tid & 15;
    // rows past B are clamped to a valid row: the loads stay unconditional (a predicated load makes hipcc branch
    // around it and drain the load queue at every join); their results are never written back.
    size_t xoff[NA], hoff[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int bb = min(r0 + row16 + 16 * i, a.B - 1);
        xoff[i] = (size_t)bb * a.KX + c4 * 4;
        hoff[i] = (size_t)a.gather_ix[bb] * slot + (size_t)bb * a.H + c4 * 4;
    }
    const float *w0p = a.wcat + (size_t)(n0 + row16) * K + c4 * 4;
    const float *w1p = w0p + (size_t)16 * K;
    // register ring: D = 4 chunks of loads in flight, as named native vectors (indexed arrays / structs of
    // HIP float4 are not promoted to registers by the compiler and end up in scratch)
    f32x4 s0a0, s0a1, s0a2, s0a3, s0b0, s0b1, s1a0, s1a1, s1a2, s1a3, s1b0, s1b1;
    f32x4 s2a0, s2a1, s2a2, s2a3, s2b0, s2b1, s3a0, s3a1, s3a2, s3a3, s3b0, s3b1;
#define LZ_LSTM_LOAD(c, q)                                                                                    \
    do {                                                                                                      \
        const int k0_ = (c) * KC;                                                                             \
        const bool inx_ = k0_ < a.KX;                                                                         \
        const float *base_ = inx_ ? a.x + k0_ : a.h_pool + (k0_ - a.KX);                                      \
        q##a0 = *reinterpret_cast<const f32x4 *>(base_ + (inx_ ? xoff[0] : hoff[0]));                         \
        q##a1 = *reinterpret_cast<const f32x4 *>(base_ + (inx_ ? xoff[1] : hoff[1]));                         \
        if constexpr (NA > 2) {                                                                               \
            q##a2 = *reinterpret_cast<const f32x4 *>(base_ + (inx_ ? xoff[NA > 2 ? 2 : 0] : hoff[NA > 2 ? 2 : 0])); \
            q##a3 = *reinterpret_cast<const f32x4 *>(base_ + (inx_ ? xoff[NA > 2 ? 3 : 0] : hoff[NA > 2 ? 3 : 0])); \
        }                                                                                                     \
        q##b0 = *reinterpret_cast<const f32x4 *>(w0p + k0_);                                                  \
        q##b1 = *reinterpret_cast<const f32x4 *>(w1p + k0_);                                                  \
    } while (0)
#define LZ_LSTM_STORE(buf, q)                                                                                 \
    do {                                                                                                      \
        *reinterpret_cast<f32x4 *>(sA(buf) + (row16 + 0) * PS + c4 * 4) = q##a0;                              \
        *reinterpret_cast<f32x4 *>(sA(buf) + (row16 + 16) * PS + c4 * 4) = q##a1;                             \
        if constexpr (NA > 2) {                                                                               \
            *reinterpret_cast<f32x4 *>(sA(buf) + (row16 + 32) * PS + c4 * 4) = q##a2;                         \
            *reinterpret_cast<f32x4 *>(sA(buf) + (row16 + 48) * PS + c4 * 4) = q##a3;                         \
        }                                                                                                     \
        *reinterpret_cast<f32x4 *>(sB(buf) + (row16 + 0) * PS + c4 * 4) = q##b0;                              \
        *reinterpret_cast<f32x4 *>(sB(buf) + (row16 + 16) * PS + c4 * 4) = q##b1;                             \
    } while (0)

    f32x4 acc[NA][2];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int coff = wv * 16 + (lane >> 4) * 4;
    // prologue: D chunks in flight, chunk 0 into LDS
    LZ_LSTM_LOAD(0, s0);
    if constexpr (1 < nchunk) LZ_LSTM_LOAD(1, s1);
    if constexpr (2 < nchunk) LZ_LSTM_LOAD(2, s2);
    if constexpr (3 < nchunk) LZ_LSTM_LOAD(3, s3);
    __builtin_amdgcn_sched_barrier(0);
    LZ_LSTM_STORE(0, s0);
    __syncthreads();
    // chunk c: its ring slot CUR is free (already in LDS) -> refill it with chunk c + D; MFMAs from LDS buffer c & 1;
    // then chunk c + 1 moves from slot NXT to the other LDS buffer.  sched_barrier keeps the refill ahead of the
    // MFMAs (the scheduler would otherwise sink the loads next to their use and expose the L2 latency).
#define LZ_LSTM_STEP(c, CUR, NXT)                                                                             \
    if constexpr ((c) < nchunk) {                                                                             \
        constexpr int buf_ = (c) & 1;                                                                         \
        if constexpr ((c) + D < nchunk) LZ_LSTM_LOAD((c) + D, CUR);                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        float4 bf_[2], af_[NA];                                                                               \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                         \
            bf_[j] = *reinterpret_cast<const float4 *>(sB(buf_) + (j * 16 + (lane & 15)) * PS + coff);        \
        _Pragma("unroll") for (int i = 0; i < NA; ++i)                                                        \
            af_[i] = *reinterpret_cast<const float4 *>(sA(buf_) + (i * 16 + (lane & 15)) * PS + coff);        \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                         \
            _Pragma("unroll") for (int i = 0; i < NA; ++i)                                                    \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                 \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(vget(af_[i], q), vget(bf_[j], q), acc[i][j], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if constexpr ((c) + 1 < nchunk) LZ_LSTM_STORE(buf_ ^ 1, NXT);                                         \
        __syncthreads();                                                                                      \
    }
    LZ_LSTM_STEP(0, s0, s1) LZ_LSTM_STEP(1, s1, s2) LZ_LSTM_STEP(2, s2, s3) LZ_LSTM_STEP(3, s3, s0)
    LZ_LSTM_STEP(4, s0, s1) LZ_LSTM_STEP(5, s1, s2) LZ_LSTM_STEP(6, s2, s3) LZ_LSTM_STEP(7, s3, s0)
    LZ_LSTM_STEP(8, s0, s1) LZ_LSTM_STEP(9, s1, s2) LZ_LSTM_STEP(10, s2, s3) LZ_LSTM_STEP(11, s3, s0)
    LZ_LSTM_STEP(12, s0, s1) LZ_LSTM_STEP(13, s1, s2) LZ_LSTM_STEP(14, s2, s3) LZ_LSTM_STEP(15, s3, s0)
    LZ_LSTM_STEP(16, s0, s1) LZ_LSTM_STEP(17, s1, s2) LZ_LSTM_STEP(18, s2, s3) LZ_LSTM_STEP(19, s3, s0)
    static_assert(NCHUNK <= 20, "extend the LZ_LSTM_STEP list");
#undef LZ_LSTM_STEP
#undef LZ_LSTM_LOAD
#undef LZ_LSTM_STORE
    // ---- split-K reduction, then the LSTM cell for MROWS rows x 8 units
    float *red = smem;                            // [4][NA*2][4][64]
    float *csum = smem + 4 * NA * 2 * 4 * 64;     // [MROWS][33]
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((wv * NA * 2 + i * 2 + j) * 4 + r) * 64 + lane] = acc[i][j][r];
    __syncthreads();
    {
        const int l = tid & 63, r = tid >> 6;
#pragma unroll
        for (int tile = 0; tile < NA * 2; ++tile) {
            const float v = red[((0 * NA * 2 + tile) * 4 + r) * 64 + l] + red[((1 * NA * 2 + tile) * 4 + r) * 64 + l] +
                            red[((2 * NA * 2 + tile) * 4 + r) * 64 + l] + red[((3 * NA * 2 + tile) * 4 + r) * 64 + l];
            const int row = (tile >> 1) * 16 + 4 * (l >> 4) + r, col = (tile & 1) * 16 + (l & 15);
            csum[row * 33 + col] = v;
        }
    }
    __syncthreads();
    for (int item = tid; item < MROWS * 8; item += 256) {
        const int row = item >> 3, u = item & 7;
        const int b = r0 + row;
        if (b >= a.B) continue;
        const int unit = (n0 >> 2) + u;
        const float gi = csum[row * 33 + 4 * u + 0] + a.bias[n0 + 4 * u + 0];
        const float gf = csum[row * 33 + 4 * u + 1] + a.bias[n0 + 4 * u + 1];
        const float gg = csum[row * 33 + 4 * u + 2] + a.bias[n0 + 4 * u + 2];
        const float go = csum[row * 33 + 4 * u + 3] + a.bias[n0 + 4 * u + 3];
        const float c_prev = a.c_pool[(size_t)a.gather_ix[b] * slot + (size_t)b * a.H + unit];
        const float cn = sigmoidf_(gf) * c_prev + sigmoidf_(gi) * tanhf_(gg);
        const float hn = sigmoidf_(go) * tanhf_(cn);
        bool reset = false;
        if (a.search_len && a.horizon > 0) reset = (a.search_len[b] % a.horizon) == 0;  // mcts_ctree.py:859-863
        a.h_out[(size_t)b * a.H + unit] = reset ? 0.0f : hn;
        a.c_out[(size_t)b * a.H + unit] = reset ? 0.0f : cn;
        // conv model: BatchNorm1d + ReLU feed the value-prefix head; MLP models (efficientzero_model_mlp.py) feed it h' itself
        a.hbn_out[(size_t)b * a.H + unit] = a.bn_scale ? fmaxf(hn * a.bn_scale[unit] + a.bn_shift[unit], 0.0f) : hn;
    }
}

// ------------------------------------------------------------------------------------------------
// LSTM step, barrier-free variant.  One workgroup = 32 rows x 16 hidden units (x 4 gates: wave g owns gate g).  The 32
// [x | h] rows are staged ONCE in LDS (32 x K floats, <= 140 KB), the weights stream from L2 straight into registers in
// MFMA-fragment order through a 12-deep ring, so the K loop is ds_read + MFMA only (no per-chunk barrier, no LDS
// stores) -- the structure of k_chain's inner loop.  grid = (H/16, ceil(B/32)): blockIdx.x walks the column tiles so
// that the workgroups sharing a weight slice sit on the same XCD (block id % 8).
// ------------------------------------------------------------------------------------------------
// XV = float4 per lane of the optional input transform (8 lanes per row): KX = 32 * XV; 0 = no transform compiled in
// MR = rows per workgroup: 32, or 16 when 32 staged rows would not fit the 160 KB of LDS (K = 1536: 64x64 observations)
// KXB > 0 (= KX / 16, compile-time): the x columns are staged first and the MFMAs over them start while the h columns -- requested
// in the same burst, parked in registers -- are still in flight; they go to LDS behind a second barrier.  One exposed round trip
// instead of two, and the second one under 36 K-steps of matrix work.
// (Measured and dropped: one 8-wave 32-row workgroup per CU -- (gate, row tile) waves sharing every weight fragment through L1,
// 411 KB instead of 684 KB through the CU -- 13.9 us against 13.6 us for two 4-wave 16-row workgroups: the kernel is bound by
// the matrix pipe (17.4 k MFMA cycles per SIMD) plus its serial staging / epilogue, not by the L1 fill rate.)
// row pitch of the staged [x | h] rows = K + 8 floats = 4 NKB + 2 bank quads: the A fragments are ds_read_b128 of 16 rows x 4 k groups, served in
// the 16-lane groups {0-3, 12-15, 20-27}, ... (MI355X_MICROARCH.md, LDS) -- a pitch of 2 (mod 4) quads puts each group on 16 distinct bank
// quads; K + 4 (1 mod 4 quads) was a 2-way conflict on every read (SQ_LDS_BANK_CONFLICT: 43 % of the LDS cycles of this kernel)
constexpr int LSTM_PAD = 8;
// SHK (split heads): columns of the combined 1x1-conv output rows per unit tile = 2 x 16 x HW / 32: 36 (6x6 latent) | 64 (8x8 latent, round 6)
// OVL (with KXB > 0; round 6, the 8x8 latent's K = 1024 + 512): the h columns take the x columns' PLACE in LDS once the x part's products are
// done (one more barrier) -- 16 rows x 1032 floats = 66 KB instead of 99 KB, so TWO workgroups fit a CU as on the 6x6 latent and one's
// staging / cell epilogue runs under the other's matrix work (99 KB: one four-wave workgroup per CU, every phase exposed)
// The launch's body as a device function (see chain_s3_body): k_lstm2 calls it with tile = blockIdx.x, r0 = blockIdx.y MR, tid = threadIdx.x;
// k_search_resident calls it from BOTH 256-thread halves of its 512-thread workgroups (two unit tiles of the group's row tile, each half
// with 70 KB of LDS of its own; every __syncthreads() in here is executed by both halves alike).  RES: the rows, pool states and gather
// indices were stored by other workgroups of the SAME launch -> sc1 loads; outputs as plain stores; `wait_inputs` is called behind the
// weight ring's requests -- the resident form waits there for the group's chain phase.
template <int NKB, int XV, int MR, int KXB, bool SH, bool GELU, int SHK, bool OVL, bool RES, class WAIT>
__device__ __forceinline__ void lstm2_body(const lz_lstm_args &a, const int tile, const int r0, const int tid, float *smem, const int ntiles,
                                           const lz_res_sim &rs, WAIT wait_inputs)
{
    static_assert(!RES || (KXB > 0 && XV == 0 && !OVL), "the resident form is written for the split-staging instance of the 6x6 latent");
    static_assert(!SH || MR == 16, "the split-head partials are written for 16-row workgroups");
    static_assert(!OVL || (KXB > 0 && MR == 16 && 2 * KXB >= NKB), "the overlay needs split staging and an x part at least as wide as the h part");
    static_assert(SHK == 36 || SHK == 64, "slice widths with a weight layout (finalize_conv_layouts)");
    constexpr int SHC4 = SHK / 4, SHB4 = (SHC4 + 3) / 4, SHP = SHK + 4;   // float4 per row slice; float4 of B operands per lane (9 -> 12 | 16 floats); LDS pitch
    constexpr int K = NKB * 16, PS = (OVL ? KXB * 16 : K) + LSTM_PAD, R = 12;
    constexpr int PSH = OVL ? (NKB - KXB) * 16 + LSTM_PAD : PS, HOFF = OVL ? 0 : KXB * 16;   // pitch / first column of the h columns' place
    constexpr int NTHR = 256, NQ = MR * 16 / NTHR, TPR = NTHR / MR;   // (row, unit) pairs per thread in the epilogue; threads staging one row
    constexpr bool TWO = MR == 32;                         // a wave computes both 16-row tiles of a 32-row workgroup
    static_assert(MR == 32 || (MR == 16 && XV == 0), "the input transform is written for 8 staging lanes per row");
    const int lane = tid & 63, wv = tid >> 6;   // smem: [MR][PS]; reused for the gate exchange
    constexpr int mt = 0;
    const int H = a.H, KX = a.KX;
    const size_t slot = (size_t)a.B * H;
    // weight ring first: its L2 round trip overlaps the staging
#ifdef LZ_DEBUG_KNOBS
    // timing experiment (debug build, LZ_DEBUG_LSTM_HOTW=1; results are then wrong): every step re-reads the first 12 fragments, i.e.
    // the weight stream comes from L1 instead of L2 -- how much of the launch is the L2 stream?
    const int wmul = (a.debug_hot_weights & 1) ? 0 : 1;
#else
    constexpr int wmul = 1;
#endif
    const float4 *wp = reinterpret_cast<const float4 *>(a.wf) + ((size_t)(tile * 4 + wv) * NKB) * 64 + lane;
    float4 wq[R];
#pragma unroll
    for (int s = 0; s < R; ++s) wq[s] = wp[(size_t)min(s, NKB - 1) * 64];
    wait_inputs();
    float *h_out = a.h_out, *c_out = a.c_out;
    if constexpr (RES) { h_out += (size_t)rs.ds * (size_t)rs.hc_step; c_out += (size_t)rs.ds * (size_t)rs.hc_step; }
    // split heads: operands of the first-layer partial products (consumed after the K loop).  value | policy heads: A = rows r0 .. r0 + 15
    // of the combined 1x1-conv outputs, columns 36 tile .. + 35 (9 k-steps of 4), B = this wave's 16 of the 64 hidden columns;
    // value-prefix head (waves 0, 1): B = units 16 tile .. + 15 x 16 hidden columns.  Requested AFTER the row staging loads below
    // (a wave's loads return in order: in front of them they would delay the first MFMA by their own -- scattered -- round trip).
    f32x4 sh_av = {0.f, 0.f, 0.f, 0.f}, sh_bv[SHB4], sh_brv;
    auto sh_request = [&]() {
        // A: 16 rows x SHK floats (contiguous per row) = 16 SHC4 float4, one per thread 0..143 | 0..255; B: this lane's 9 (+ 3 pad) | 16 weights, | 4
        const int row = min(tid / SHC4, 15), c4 = tid % SHC4;
        const int bb = min(r0 + row, a.B - 1);
#ifdef LZ_DEBUG_KNOBS
        if (a.debug_hot_weights & 2) { for (int i = 0; i < SHB4; ++i) sh_bv[i] = sh_av; sh_brv = sh_av; return; }
#endif
        if constexpr (RES) sh_av = load_sc1_f4(a.sh_pv, (size_t)bb * a.sh_kc + SHK * tile + 4 * c4);
        else sh_av = *reinterpret_cast<const f32x4 *>(a.sh_pv + (size_t)bb * a.sh_kc + SHK * tile + 4 * c4);
        const float *bp = a.sh_w1c + (((size_t)tile * 4 + wv) * 64 + lane) * (4 * SHB4);
#pragma unroll
        for (int i = 0; i < SHB4; ++i) sh_bv[i] = *reinterpret_cast<const f32x4 *>(bp + 4 * i);
        sh_brv = *reinterpret_cast<const f32x4 *>(a.sh_w1r + (((size_t)tile * 2 + (wv & 1)) * 64 + lane) * 4);
    };
    // everything the cell epilogue needs for this thread's two (row, unit) pairs is requested now: previous cell state, gate
    // biases, BatchNorm scale / shift, reset flag (unconditional loads; dummies where a pointer is null)
    float c_prev[NQ], gb[NQ][4], bns[NQ], bnt[NQ];
    int slen[NQ];
    const float *bnsp = a.bn_scale ? a.bn_scale : a.bias, *bntp = a.bn_scale ? a.bn_shift : a.bias;
    const int32_t *slp = a.search_len ? a.search_len : a.gather_ix;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int p = tid + NTHR * q, row = p >> 4, u = p & 15;
        const int b = min(r0 + row, a.B - 1), unit = tile * 16 + u;
        if constexpr (RES) c_prev[q] = load_sc1_f(a.c_pool + (size_t)load_sc1_i(a.gather_ix + b) * slot + (size_t)b * H + unit);
        else c_prev[q] = a.c_pool[(size_t)a.gather_ix[b] * slot + (size_t)b * H + unit];
        const float4 b4 = *reinterpret_cast<const float4 *>(a.bias + 4 * unit);
        gb[q][0] = b4.x; gb[q][1] = b4.y; gb[q][2] = b4.z; gb[q][3] = b4.w;
        bns[q] = bnsp[unit]; bnt[q] = bntp[unit];
        slen[q] = RES ? load_sc1_i(slp + b) : slp[b];
    }
    // stage the rows: [x (KX) | h (H)] per row; TPR (8) threads per row, batches of 12 float4 loads in flight per thread
    constexpr int K4 = K / 4, NI = (K4 + TPR - 1) / TPR, NBATCH = 12;
    constexpr int NXS = KXB > 0 ? KXB * 4 / TPR : 0, NHS = KXB > 0 ? (K4 - KXB * 4) / TPR : 1;
    static_assert(KXB == 0 || (XV == 0 && (KXB * 4) % TPR == 0 && (K4 - KXB * 4) % TPR == 0), "split staging needs whole thread strides");
    f32x4 hv[NHS];  // native vectors: an array of HIP's float4 struct that lives across the K loop ends up in scratch
    if constexpr (KXB > 0) {
        const int row = tid / TPR, part = tid % TPR;
        const int b = min(r0 + row, a.B - 1);
        const float *xrow = a.x + (size_t)b * KX;
        const size_t hoff = (size_t)(RES ? load_sc1_i(a.gather_ix + b) : a.gather_ix[b]) * slot + (size_t)b * H;
        const float *hrow = a.h_pool + hoff;
        float *dst = smem + row * PS;
        f32x4 xv[NXS];
#pragma unroll
        for (int i = 0; i < NXS; ++i) {
            if constexpr (RES) xv[i] = load_sc1_f4(a.x, (size_t)b * KX + (part + TPR * i) * 4);
            else xv[i] = *reinterpret_cast<const f32x4 *>(xrow + (part + TPR * i) * 4);
        }
#pragma unroll
        for (int i = 0; i < NHS; ++i) {
            if constexpr (RES) hv[i] = load_sc1_f4(a.h_pool, hoff + (part + TPR * i) * 4);
            else hv[i] = *reinterpret_cast<const f32x4 *>(hrow + (part + TPR * i) * 4);
        }
        if constexpr (SH) sh_request();
#pragma unroll
        for (int i = 0; i < NXS; ++i) *reinterpret_cast<f32x4 *>(dst + (part + TPR * i) * 4) = xv[i];
    } else {
        const int row = tid / TPR, part = tid % TPR;
        const int b = min(r0 + row, a.B - 1);
        const int kx4 = KX >> 2;
        const float *xrow = a.x + (size_t)b * KX;
        const float *hrow = a.h_pool + (size_t)a.gather_ix[b] * slot + (size_t)b * H - KX;
        float *dst = smem + row * PS;
#pragma unroll
        for (int i0 = 0; i0 < NI; i0 += NBATCH) {
            float4 v[NBATCH];
#pragma unroll
            for (int i = 0; i < NBATCH; ++i) {
                const int k4 = min(part + TPR * (i0 + i), K4 - 1);
                v[i] = *reinterpret_cast<const float4 *>((k4 < kx4 ? xrow : hrow) + k4 * 4);
            }
#pragma unroll
            for (int i = 0; i < NBATCH; ++i) {
                const int k4 = part + TPR * (i0 + i);
                if (i0 + i < NI && k4 < K4) *reinterpret_cast<float4 *>(dst + k4 * 4) = v[i];
            }
        }
        if constexpr (SH) sh_request();   // (behind the row staging loads: in front of them it would delay the first MFMA by its own round trip)
        if constexpr (XV > 0) {
            // x's producer left its LayerNorm / activation to us (vector-observation models): the 8 lanes that staged the row
            // finish it in place; float4 granularity, straight-line, every load issued before the first use
            const bool has_ln = a.x_ln_g != nullptr;
            const float *gp = has_ln ? a.x_ln_g : a.bias, *bp = has_ln ? a.x_ln_b : a.bias;  // dummies are valid memory
            float4 xv[XV], gv[XV], bv[XV];
#pragma unroll
            for (int i = 0; i < XV; ++i) {
                const int k = 4 * (part + 8 * i);
                xv[i] = *reinterpret_cast<const float4 *>(dst + k);
                gv[i] = *reinterpret_cast<const float4 *>(gp + k);
                bv[i] = *reinterpret_cast<const float4 *>(bp + k);
            }
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < XV; ++i) sum += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w);
            sum = group_sum<8>(sum);  // the 8 lanes that staged the row
            const float mean_ln = sum / (float)KX;
            float sq = 0.0f;
#pragma unroll
            for (int i = 0; i < XV; ++i) {
                const float d0 = xv[i].x - mean_ln, d1 = xv[i].y - mean_ln, d2 = xv[i].z - mean_ln, d3 = xv[i].w - mean_ln;
                sq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
            sq = group_sum<8>(sq);
            const float mean = has_ln ? mean_ln : 0.0f, rstd = has_ln ? 1.0f / sqrtf(sq / (float)KX + a.x_ln_eps) : 1.0f;
            auto fin = [&](float u, float g, float be) {
                u = (u - mean) * rstd;
                u = has_ln ? u * g + be : u;
                const float y = 0.7978845608028654f * (u + 0.044715f * u * u * u);
                const float gl = 0.5f * u * (2.0f - 2.0f * __frcp_rn(1.0f + __expf(2.0f * y)));  // GELU(tanh), hardware exp / rcp
                return a.x_act == 1 ? fmaxf(u, 0.0f) : (a.x_act == 2 ? gl : u);
            };
#pragma unroll
            for (int i = 0; i < XV; ++i) {
                *reinterpret_cast<float4 *>(dst + 4 * (part + 8 * i)) =
                    make_float4(fin(xv[i].x, gv[i].x, bv[i].x), fin(xv[i].y, gv[i].y, bv[i].y), fin(xv[i].z, gv[i].z, bv[i].z), fin(xv[i].w, gv[i].w, bv[i].w));
            }
        }
    }
    __syncthreads();
    const float *sA0 = smem + (mt * 16 + (lane & 15)) * PS + (lane >> 4) * 4;
    const float *sA1 = sA0 + (TWO ? 16 : 0) * PS;  // only a four-wave 32-row workgroup computes a second tile per wave
    const float *sAh = OVL ? smem + (mt * 16 + (lane & 15)) * PSH + (lane >> 4) * 4 - KXB * 16 : sA0;   // (OVL: k step s >= KXB reads sAh + s * 16)
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    float4 a0 = *reinterpret_cast<const float4 *>(sA0), a1 = *reinterpret_cast<const float4 *>(sA1);
#pragma unroll
    for (int s = 0; s < NKB; ++s) {
        if constexpr (KXB > 0) {
            if (s == KXB) {  // the h columns have arrived behind the x part's MFMAs: into LDS, then on
                const int row = tid / TPR, part = tid % TPR;
                if constexpr (OVL) __syncthreads();   // every wave has read its last x fragment: their place is free
                float *dst = smem + row * PSH + HOFF;
#pragma unroll
                for (int i = 0; i < NHS; ++i) *reinterpret_cast<f32x4 *>(dst + (part + TPR * i) * 4) = hv[i];
                __syncthreads();
                a0 = *reinterpret_cast<const float4 *>((OVL ? sAh : sA0) + s * 16);
                a1 = *reinterpret_cast<const float4 *>(sA1 + s * 16);
            }
        }
        const float4 bfr = wq[s % R];
        if (s + R < NKB) wq[s % R] = wp[(size_t)((s + R) * wmul + (s % R) * (1 - wmul)) * 64];
        float4 n0 = a0, n1 = a1;
        if (s + 1 < NKB && (KXB == 0 || s + 1 != KXB)) {
            n0 = *reinterpret_cast<const float4 *>(((OVL && s + 1 > KXB) ? sAh : sA0) + (s + 1) * 16);
            n1 = *reinterpret_cast<const float4 *>(sA1 + (s + 1) * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vget(a0, j), vget(bfr, j), acc0, 0, 0, 0);
            if constexpr (TWO) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vget(a1, j), vget(bfr, j), acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        a0 = n0;
        a1 = n1;
    }
    __syncthreads();  // every wave is done reading the staged rows: the buffer becomes the gate exchange [4][32][17]
    float *sG = smem;
    float *sHb = smem + 4 * MR * 17;   // split heads: relu(bn(h')) of this workgroup's 16 rows x 16 units [16][17]
    float *sA2 = sHb + 16 * 17;        //              the value | policy heads' input slice [16 rows][SHK + 4]
    float *sP = sA2 + 16 * SHP;        //              the partial block of this workgroup [16 rows][3 heads][32 hidden]
    if constexpr (SH) {
        if (tid < 16 * SHC4) *reinterpret_cast<f32x4 *>(sA2 + (tid / SHC4) * SHP + (tid % SHC4) * 4) = sh_av;
    }
    {
        const int col = lane & 15, rq = 4 * (lane >> 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sG[(wv * MR + mt * 16 + rq + q) * 17 + col] = acc0[q];
            if constexpr (TWO) sG[(wv * MR + 16 + rq + q) * 17 + col] = acc1[q];
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int p = tid + NTHR * q, row = p >> 4, u = p & 15;
        const int b = r0 + row;
        if (b >= a.B) { if constexpr (SH) sHb[row * 17 + u] = 0.0f; continue; }
        const int unit = tile * 16 + u;
        const float gi = sG[(0 * MR + row) * 17 + u] + gb[q][0];
        const float gf = sG[(1 * MR + row) * 17 + u] + gb[q][1];
        const float gg = sG[(2 * MR + row) * 17 + u] + gb[q][2];
        const float go = sG[(3 * MR + row) * 17 + u] + gb[q][3];
        const float cn = sigmoidf_(gf) * c_prev[q] + sigmoidf_(gi) * tanhf_(gg);
        const float hn = sigmoidf_(go) * tanhf_(cn);
        const bool reset = a.search_len && a.horizon > 0 && (slen[q] % a.horizon) == 0;  // mcts_ctree.py:859-863
        const float hb = a.bn_scale ? act_<GELU>(hn * bns[q] + bnt[q]) : hn;
        if constexpr (RES) {   // plain stores: later simulations of this launch read them from this XCD's L2
            h_out[(size_t)b * H + unit] = reset ? 0.0f : hn;
            c_out[(size_t)b * H + unit] = reset ? 0.0f : cn;
            a.hbn_out[(size_t)b * H + unit] = hb;
        } else {
            store_wt(h_out + (size_t)b * H + unit, reset ? 0.0f : hn);
            store_wt(c_out + (size_t)b * H + unit, reset ? 0.0f : cn);
            store_wt(a.hbn_out + (size_t)b * H + unit, hb);
        }
        if constexpr (SH) sHb[row * 17 + u] = hb;
    }
    if constexpr (SH) {
        // value | policy heads, first layer: [16 rows x 36] x [36 x this wave's 16 of the 64 hidden columns] (sA2 was written before the
        // gate barrier); D[row = 4 (lane >> 4) + q][col = lane & 15] -> sP[row][head = wave / 2][16 (wave & 1) + col]
        {
            f32x4 pacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < SHC4; ++ks)
                pacc = __builtin_amdgcn_mfma_f32_16x16x4f32(sA2[(lane & 15) * SHP + 4 * ks + (lane >> 4)], sh_bv[ks >> 2][ks & 3], pacc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) sP[(4 * (lane >> 4) + q) * 96 + (wv >> 1) * 32 + 16 * (wv & 1) + (lane & 15)] = pacc[q];
        }
        __syncthreads();   // relu(bn(h')) of all 16 x 16 (row, unit) pairs is in sHb
        if (wv < 2) {      // value-prefix head, first layer: [16 rows x 16 units of this tile] x [16 x 16 hidden columns of wave 0 | 1]
            f32x4 pacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                pacc = __builtin_amdgcn_mfma_f32_16x16x4f32(sHb[(lane & 15) * 17 + 4 * ks + (lane >> 4)], sh_brv[ks], pacc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) sP[(4 * (lane >> 4) + q) * 96 + 64 + 16 * wv + (lane & 15)] = pacc[q];
        }
        __syncthreads();
        // the block goes out as whole 128-byte lines: (row, head) = 32 floats at [root][head][unit tile][32]
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + NTHR * i;
            if (idx < 16 * 24) {
                const int row = idx / 24, c4 = idx % 24;
#ifdef LZ_DEBUG_KNOBS
                if (a.debug_hot_weights & 4) continue;
#endif
                if (r0 + row < a.B) {
                    float *pd = a.sh_part + ((size_t)(r0 + row) * 3 + (c4 >> 3)) * (ntiles * 32) + (size_t)tile * 32 + (c4 & 7) * 4;
                    if constexpr (RES) *reinterpret_cast<f32x4 *>(pd) = *reinterpret_cast<const f32x4 *>(sP + row * 96 + c4 * 4);
                    else store_wt(pd, *reinterpret_cast<const f32x4 *>(sP + row * 96 + c4 * 4));
                }
            }
        }
    }
}

template <int NKB, int XV = 0, int MR = 32, int KXB = 0, bool SH = false, bool GELU = false, int SHK = 36, bool OVL = false>
__global__ __launch_bounds__(256) void k_lstm2(lz_lstm_args a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    lz_stamp_begin(a.stamp);
    lstm2_body<NKB, XV, MR, KXB, SH, GELU, SHK, OVL, false>(a, (int)blockIdx.x, (int)blockIdx.y * MR, (int)threadIdx.x, smem, (int)gridDim.x, lz_res_sim{}, [] {});
    lz_stamp_end(a.stamp);
}

// ------------------------------------------------------------------------------------------------
// ONE LAUNCH PER SIMULATION (round 6; opt-in: LZ_SIM_ONE_LAUNCH=1; EfficientZero, 6x6 latent, split heads, 128 | 256 roots): the LSTM launch of
// simulation s - 1 and the tree-fused chain launch of simulation s -- k_lstm2<68,0,16,36,true> and k_chain_s3<6,6,1,true> -- as the two
// PHASES of one launch.  Of a 48.9 us simulation 5.3 us are the two kernel boundaries (execution end -> the next launch's first workgroup)
// plus two launch ramps; VERDICT r5 #3 asked for the resident form (one launch per SEARCH), whose hand-off this uses:
//   * one workgroup per root and CU (512 threads, the chain's 153 KB of LDS); in the LSTM phase its two 256-thread halves are two unit
//     tiles of the group's 16-row tile (2 x 70 KB of LDS);
//   * GROUPS of 16 roots (= one LSTM row tile = the producers of a root's head partials) are formed at run time from HW_REG_XCC_ID, so that
//     the hand-off stays inside one XCD's L2: producer = plain stores -> s_waitcnt vmcnt(0) -> barrier -> one relaxed agent-scope atomic
//     add on the group's counter of this launch; consumer = one lane polls it (sc1 load: L2-served) -> barrier -> the partials by sc1 loads
//     (heads_in_prologue<RES>).  No buffer_wbl2, no buffer_inv: nothing is written back or invalidated.  Everything else a phase reads
//     was written by an EARLIER launch (rows, pool states, the tree) or by its own workgroup;
//   * why not the loop over all simulations: inside a loop the compiler hoists both bodies' loop-invariant argument loads and keeps
//     them live across the other body -- 256 registers + 1.7 KB of scratch per lane (measured; the chain body alone is 244 registers);
//     straight-line, this kernel is 245 registers and no scratch.  The boundary that remains per simulation is the one behind the chain phase;
//   * every spin is BOUNDED: a wait that runs out (a workgroup that is not resident, an XCD with another share of the workgroups) raises
//     ctl->fault, lz_search reports LZ_ERR_STATE and the caller repeats the env-step on the two-launch path.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool res_wait(const unsigned *f, unsigned want, unsigned *fault)
{
    int spins = 0;
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 18) || __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_fetch_add(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}

__global__ __launch_bounds__(512) void k_sim_fused(lz_lstm_args la, lz_chain_args ca, lz_tree_step step, lz_resident_args ra)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // (no static LDS: with any, hipFuncSetAttribute(MaxDynamicSharedMemorySize = 160 KB) is an invalid argument.)  Four words behind the chain's layout
    constexpr int LDS_CHAIN = (int)(LZ_S3_LDS / 4);   // floats: k_chain_s3<6,6>'s
    int *s_ids = reinterpret_cast<int *>(smem + LDS_CHAIN);
    const int tid = threadIdx.x;
    lz_res_ctl *ctl = ra.ctl;
    const int per_xcd = ra.B >> 3;                  // workgroups of an XCD (the dispatcher deals them round-robin: checked, not assumed)
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xf;
        // tickets are monotonic over the launches of a search (the control block is zeroed once per search): launch k hands out k per_xcd ...
        const int slot = (int)atomicAdd(&ctl->xcc_count[xcc], 1u) - ra.launch * per_xcd;
        const bool ok = slot >= 0 && slot < per_xcd && xcc < 8;
        if (!ok) __hip_atomic_fetch_add(&ctl->fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int sl = ok ? slot : 0;
        s_ids[0] = (int)(xcc & 7) * (per_xcd >> 4) + (sl >> 4);
        s_ids[1] = sl & 15;
    }
    __syncthreads();
    const int grp = __builtin_amdgcn_readfirstlane(s_ids[0]), mem = __builtin_amdgcn_readfirstlane(s_ids[1]);
    const int b = grp * 16 + mem, half = tid >> 8;
    unsigned *flag = ctl->flags + (size_t)ra.launch * ra.ngroups + grp;
    constexpr int LDS_HALF = 16 * (68 * 16 + LSTM_PAD);   // floats of one half's staged rows
    lz_res_sim rs;
    rs.ds = 0; rs.B = ra.B; rs.BA = ra.BA; rs.lat_step = 0; rs.hc_step = 0;
    // ---- phase 1: the LSTM step of the previous simulation for the group's 16 rows, unit tiles 2 mem and 2 mem + 1 (+ the head partials)
    // (its inputs were written by earlier launches: plain loads; its outputs stay write-through -- the pool rows are read by later launches,
    // and a partial block that an sc1 store dropped from L2 costs the head waves' sc1 loads a memory round trip, not a stale line)
    lstm2_body<68, 0, 16, 36, true, false, 36, false, false>(la, 2 * mem + half, grp * 16, tid & 255, smem + half * LDS_HALF, 32, rs, [] {});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ids[2] = res_wait(flag, 16u, &ctl->fault) ? 1 : 0;      // the whole group's partials are in this XCD's L2
    }
    __syncthreads();
    if (s_ids[2] == 0) return;
    // ---- phase 2: this simulation's chain launch for root b (heads of the previous leaf + tree step in the prologue)
    chain_s3_body<6, 6, 1, true, 1>(ca, step, b, ra.B, rs);
}

// ------------------------------------------------------------------------------------------------
// FAST MODE (lz_model_cfg::precision = 1): k_lstm2<68, 0, 16, 36, SH> with the gate product on v_mfma_f32_16x16x32_bf16 -- the rows
// [x | h] are rounded to bf16 while they are staged ([16][K + 16] bf16 in LDS: one ds_read_b128 per A fragment), the gate weights are bf16
// fragments [H/16][4 gates][K/32][64 lanes][8] (lz_lstm_args::wb), accumulation / cell / BatchNorm / split-head partials in fp32 as in
// k_lstm2.  34 MFMAs per wave instead of 272; a workgroup streams 139 KB of weights instead of 278 KB.
// ------------------------------------------------------------------------------------------------
template <bool SH, int NKB = 68, int KXB = 36>   // 68 / 36: 16 x 36 + 512 columns (6x6 latent); 96 / 64: 16 x 64 + 512 (8x8 latent)
__global__ __launch_bounds__(256) void k_lstm_b(lz_lstm_args a)
{
    constexpr int K = NKB * 16, NS = K / 32, SX = KXB / 2, PB = K + 16, R = 12, MR = 16;   // row pitch = 2 (mod 4) bank quads: conflict-free ds_read_b128
    static_assert(!SH || NKB == 68, "the split-head partials exist for the 6x6 latent");
    constexpr int NTHR = 256, NQ = MR * 16 / NTHR, TPR = NTHR / MR;
    extern __shared__ __attribute__((aligned(16))) float smem[];  // bf16 [16][PB]; reused (fp32) for the gate exchange
    __bf16 *sR = reinterpret_cast<__bf16 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tile = blockIdx.x, r0 = blockIdx.y * MR;
    const int H = a.H, KX = a.KX;
    const size_t slot = (size_t)a.B * H;
    lz_stamp_begin(a.stamp);
    const bf16x8 *wp = reinterpret_cast<const bf16x8 *>(a.wb) + ((size_t)(tile * 4 + wv) * NS) * 64 + lane;
    bf16x8 wq[R];
#pragma unroll
    for (int s = 0; s < R; ++s) wq[s] = wp[(size_t)s * 64];
    f32x4 sh_av = {0.f, 0.f, 0.f, 0.f}, sh_bv[3], sh_brv;
    auto sh_request = [&]() {
        const int row = min(tid / 9, 15), c4 = tid % 9;
        const int bb = min(r0 + row, a.B - 1);
        sh_av = *reinterpret_cast<const f32x4 *>(a.sh_pv + (size_t)bb * a.sh_kc + 36 * tile + 4 * c4);
        const float *bp = a.sh_w1c + (((size_t)tile * 4 + wv) * 64 + lane) * 12;
#pragma unroll
        for (int i = 0; i < 3; ++i) sh_bv[i] = *reinterpret_cast<const f32x4 *>(bp + 4 * i);
        sh_brv = *reinterpret_cast<const f32x4 *>(a.sh_w1r + (((size_t)tile * 2 + (wv & 1)) * 64 + lane) * 4);
    };
    float c_prev[NQ], gb[NQ][4], bns[NQ], bnt[NQ];
    int slen[NQ];
    const float *bnsp = a.bn_scale ? a.bn_scale : a.bias, *bntp = a.bn_scale ? a.bn_shift : a.bias;
    const int32_t *slp = a.search_len ? a.search_len : a.gather_ix;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int p = tid + NTHR * q, row = p >> 4, u = p & 15;
        const int b = min(r0 + row, a.B - 1), unit = tile * 16 + u;
        c_prev[q] = a.c_pool[(size_t)a.gather_ix[b] * slot + (size_t)b * H + unit];
        const float4 b4 = *reinterpret_cast<const float4 *>(a.bias + 4 * unit);
        gb[q][0] = b4.x; gb[q][1] = b4.y; gb[q][2] = b4.z; gb[q][3] = b4.w;
        bns[q] = bnsp[unit]; bnt[q] = bntp[unit];
        slen[q] = slp[b];
    }
    // stage the rows: 16 threads per row; the x columns first, the h columns (a dependent pool gather) arrive under the x part's MFMAs
    constexpr int K4 = K / 4, NXS = KXB * 4 / TPR, NHS = (K4 - KXB * 4) / TPR;
    static_assert((KXB * 4) % TPR == 0 && (K4 - KXB * 4) % TPR == 0 && (KXB % 2) == 0, "whole thread strides, whole 32-column steps");
    f32x4 hv[NHS];
    auto put = [&](__bf16 *dst, const f32x4 &v) {
        bf16x4 h;
        h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
        *reinterpret_cast<bf16x4 *>(dst) = h;
    };
    {
        const int row = tid / TPR, part = tid % TPR;
        const int b = min(r0 + row, a.B - 1);
        const float *xrow = a.x + (size_t)b * KX;
        const float *hrow = a.h_pool + (size_t)a.gather_ix[b] * slot + (size_t)b * H;
        __bf16 *dst = sR + row * PB;
        f32x4 xv[NXS];
#pragma unroll
        for (int i = 0; i < NXS; ++i) xv[i] = *reinterpret_cast<const f32x4 *>(xrow + (part + TPR * i) * 4);
#pragma unroll
        for (int i = 0; i < NHS; ++i) hv[i] = *reinterpret_cast<const f32x4 *>(hrow + (part + TPR * i) * 4);
        if constexpr (SH) sh_request();
#pragma unroll
        for (int i = 0; i < NXS; ++i) put(dst + (part + TPR * i) * 4, xv[i]);
    }
    __syncthreads();
    const __bf16 *sA = sR + (lane & 15) * PB + (lane >> 4) * 8;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s == SX) {  // the h columns have arrived behind the x part's MFMAs: into LDS, then on
            const int row = tid / TPR, part = tid % TPR;
            __bf16 *dst = sR + row * PB + KXB * 16;
#pragma unroll
            for (int i = 0; i < NHS; ++i) put(dst + (part + TPR * i) * 4, hv[i]);
            __syncthreads();
        }
        const bf16x8 bfr = wq[s % R];
        if (s + R < NS) wq[s % R] = wp[(size_t)(s + R) * 64];
        const bf16x8 af = *reinterpret_cast<const bf16x8 *>(sA + s * 32);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc0, 0, 0, 0);
    }
    __syncthreads();  // every wave is done reading the staged rows: the buffer becomes the gate exchange [4][16][17]
    float *sG = smem;
    float *sHb = smem + 4 * MR * 17;
    float *sA2 = sHb + 16 * 17;
    float *sP = sA2 + 16 * 40;
    if constexpr (SH) {
        if (tid < 144) *reinterpret_cast<f32x4 *>(sA2 + (tid / 9) * 40 + (tid % 9) * 4) = sh_av;
    }
    {
        const int col = lane & 15, rq = 4 * (lane >> 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) sG[(wv * MR + rq + q) * 17 + col] = acc0[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int p = tid + NTHR * q, row = p >> 4, u = p & 15;
        const int b = r0 + row;
        if (b >= a.B) { if constexpr (SH) sHb[row * 17 + u] = 0.0f; continue; }
        const int unit = tile * 16 + u;
        const float gi = sG[(0 * MR + row) * 17 + u] + gb[q][0];
        const float gf = sG[(1 * MR + row) * 17 + u] + gb[q][1];
        const float gg = sG[(2 * MR + row) * 17 + u] + gb[q][2];
        const float go = sG[(3 * MR + row) * 17 + u] + gb[q][3];
        const float cn = sigmoidf_(gf) * c_prev[q] + sigmoidf_(gi) * tanhf_(gg);
        const float hn = sigmoidf_(go) * tanhf_(cn);
        const bool reset = a.search_len && a.horizon > 0 && (slen[q] % a.horizon) == 0;  // mcts_ctree.py:859-863
        store_wt(a.h_out + (size_t)b * H + unit, reset ? 0.0f : hn);
        store_wt(a.c_out + (size_t)b * H + unit, reset ? 0.0f : cn);
        const float hb = a.bn_scale ? fmaxf(hn * bns[q] + bnt[q], 0.0f) : hn;
        store_wt(a.hbn_out + (size_t)b * H + unit, hb);
        if constexpr (SH) sHb[row * 17 + u] = hb;
    }
    if constexpr (SH) {
        {
            f32x4 pacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 9; ++ks)
                pacc = __builtin_amdgcn_mfma_f32_16x16x4f32(sA2[(lane & 15) * 40 + 4 * ks + (lane >> 4)], sh_bv[ks >> 2][ks & 3], pacc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) sP[(4 * (lane >> 4) + q) * 96 + (wv >> 1) * 32 + 16 * (wv & 1) + (lane & 15)] = pacc[q];
        }
        __syncthreads();
        if (wv < 2) {
            f32x4 pacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                pacc = __builtin_amdgcn_mfma_f32_16x16x4f32(sHb[(lane & 15) * 17 + 4 * ks + (lane >> 4)], sh_brv[ks], pacc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) sP[(4 * (lane >> 4) + q) * 96 + 64 + 16 * wv + (lane & 15)] = pacc[q];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + NTHR * i;
            if (idx < 16 * 24) {
                const int row = idx / 24, c4 = idx % 24;
                if (r0 + row < a.B)
                    store_wt(a.sh_part + ((size_t)(r0 + row) * 3 + (c4 >> 3)) * (gridDim.x * 32) + (size_t)tile * 32 + (c4 & 7) * 4,
                             *reinterpret_cast<const f32x4 *>(sP + row * 96 + c4 * 4));
            }
        }
    }
    lz_stamp_end(a.stamp);
}

// ------------------------------------------------------------------------------------------------
// LSTM step, 32-row workgroups with PIPELINED staging (round 4).  What bounds k_lstm2 at its 16-row default is the vector-memory path
// of a CU: 2 x (278 KB of gate weights + 70 KB of rows) = 696 KB per CU and launch at the ~34 B/clk a CU receives under
// MFMA load.  32 rows per workgroup halve the weight stream -- every fragment feeds TWO 16-row MFMA tiles from the same registers:
// 278 + 139 = 417 KB per CU -- but the k_lstm2<..., 32> form staged all 139 KB before its first MFMA with nothing else resident on
// the CU to hide it (14.2 us against 13.6 us).  Here the rows arrive in chunks of 16 k-steps (32 rows x 256 columns = 32 KB): chunk
// c + 1 is requested into registers before the MFMAs of chunk c start and written to ITS OWN region of LDS after them (no buffer is
// reused, so one barrier per chunk and no wait on the way in); the first MFMA needs 32 KB, and that chunk is x columns only (no
// dependent pool gather).  grid = (H / 16, ceil(B / 32)) = 256 workgroups at 256 roots: one per CU, unit tiles walk the XCDs.
// Same k order and accumulation as k_lstm2 -> bit-identical gates, cell states and split-head partials (tools/dump_search.py --compare,
// 256 x 50 and 67 x 12).  MEASURED (same box, alternating, in-graph stamps): 15.15 us per launch against 15.05 us for the 16-row default
// (first workgroup start -> last end 12.9 vs 12.7 us) -- halving the weight stream buys nothing: the 9 us of matrix work plus the
// exposed prologue (first chunk + ring) and cell epilogue of ONE workgroup per CU cost what two resident 16-row workgroups lose to the
// stream.  Kept behind LZ_LSTM3=1 (qualified, not the default).
// SH: the split-head first layers (lz_lstm_args::sh_*) for both 16-row halves.
// ------------------------------------------------------------------------------------------------
template <int NKB, bool SH>
__global__ __launch_bounds__(256) void k_lstm3(lz_lstm_args a)
{
    constexpr int K = NKB * 16, PS = K + LSTM_PAD, R = 12, MR = 32, NTHR = 256;
    constexpr int CH = 16, NCH = (NKB + CH - 1) / CH;   // k-steps per staging chunk; chunks
    constexpr int K4 = K / 4, F4C = CH * 4, NLD = F4C / 8;   // float4 per row; per row and chunk; per thread and chunk (8 threads per row)
    constexpr int NQ = MR * 16 / NTHR;                  // (row, unit) pairs per thread in the cell epilogue
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [32][PS]; reused for the gate exchange
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tile = blockIdx.x, r0 = blockIdx.y * MR;
    const int H = a.H, KX = a.KX;
    const size_t slot = (size_t)a.B * H;
    lz_stamp_begin(a.stamp);
    // ---- requests in the order of use: the first weight fragments, the first chunk of rows (x columns: no gather), the pool slot
    const float4 *wp = reinterpret_cast<const float4 *>(a.wf) + ((size_t)(tile * 4 + wv) * NKB) * 64 + lane;
    float4 wq[R];
#pragma unroll
    for (int s = 0; s < R; ++s) wq[s] = wp[(size_t)min(s, NKB - 1) * 64];
    const int srow = tid >> 3, spart = tid & 7;
    const int sb = min(r0 + srow, a.B - 1);
    const int kx4 = KX >> 2;
    const float *xrow = a.x + (size_t)sb * KX;
    float *sdst = smem + srow * PS;
    f32x4 cv[NLD];
    // chunk 0 lies inside the x columns for every shape this kernel is instantiated for (KX >= 256)
#pragma unroll
    for (int i = 0; i < NLD; ++i) cv[i] = *reinterpret_cast<const f32x4 *>(xrow + (spart + 8 * i) * 4);
    const float *hrow = a.h_pool + (size_t)a.gather_ix[sb] * slot + (size_t)sb * H - KX;   // (k4 >= kx4 ? hrow : xrow) + 4 k4
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int k4 = min(c * F4C + spart + 8 * i, K4 - 1);
            cv[i] = *reinterpret_cast<const f32x4 *>((k4 < kx4 ? xrow : hrow) + k4 * 4);
        }
    };
    auto store_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int k4 = c * F4C + spart + 8 * i;
            if (k4 < K4) *reinterpret_cast<f32x4 *>(sdst + k4 * 4) = cv[i];
        }
    };
    // split heads: operands of the first-layer partial products (consumed after the K loop): rows r0 .. r0 + 31 of the combined
    // 1x1-conv outputs, columns 36 tile .. + 35 (288 float4: one per thread + 32 more), this wave's weight slices
    f32x4 sh_av[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, sh_bv[3], sh_brv;
    if constexpr (SH) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = min(tid + NTHR * u, 32 * 9 - 1), row = idx / 9, c4 = idx - row * 9;
            const int bb = min(r0 + row, a.B - 1);
            sh_av[u] = *reinterpret_cast<const f32x4 *>(a.sh_pv + (size_t)bb * a.sh_kc + 36 * tile + 4 * c4);
        }
        const float *bp = a.sh_w1c + (((size_t)tile * 4 + wv) * 64 + lane) * 12;
#pragma unroll
        for (int i = 0; i < 3; ++i) sh_bv[i] = *reinterpret_cast<const f32x4 *>(bp + 4 * i);
        sh_brv = *reinterpret_cast<const f32x4 *>(a.sh_w1r + (((size_t)tile * 2 + (wv & 1)) * 64 + lane) * 4);
    }
    // the cell epilogue's operands (previous cell state, gate biases, BatchNorm, reset flag) for this thread's two (row, unit) pairs
    float c_prev[NQ], gb[NQ][4], bns[NQ], bnt[NQ];
    int slen[NQ];
    const float *bnsp = a.bn_scale ? a.bn_scale : a.bias, *bntp = a.bn_scale ? a.bn_shift : a.bias;
    const int32_t *slp = a.search_len ? a.search_len : a.gather_ix;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int p = tid + NTHR * q, row = p >> 4, u = p & 15;
        const int b = min(r0 + row, a.B - 1), unit = tile * 16 + u;
        c_prev[q] = a.c_pool[(size_t)a.gather_ix[b] * slot + (size_t)b * H + unit];
        const float4 b4 = *reinterpret_cast<const float4 *>(a.bias + 4 * unit);
        gb[q][0] = b4.x; gb[q][1] = b4.y; gb[q][2] = b4.z; gb[q][3] = b4.w;
        bns[q] = bnsp[unit]; bnt[q] = bntp[unit];
        slen[q] = slp[b];
    }
    store_chunk(0);
    __syncthreads();
    const float *sA0 = smem + (lane & 15) * PS + (lane >> 4) * 4, *sA1 = sA0 + 16 * PS;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) load_chunk(c + 1);   // in flight under this chunk's MFMAs
        const int s0 = c * CH, s1 = (c + 1) * CH < NKB ? (c + 1) * CH : NKB;
        float4 a0 = *reinterpret_cast<const float4 *>(sA0 + s0 * 16), a1 = *reinterpret_cast<const float4 *>(sA1 + s0 * 16);
#pragma unroll
        for (int s = s0; s < s1; ++s) {
            const float4 bfr = wq[s % R];
            if (s + R < NKB) wq[s % R] = wp[(size_t)(s + R) * 64];
            float4 n0 = a0, n1 = a1;
            if (s + 1 < s1) {
                n0 = *reinterpret_cast<const float4 *>(sA0 + (s + 1) * 16);
                n1 = *reinterpret_cast<const float4 *>(sA1 + (s + 1) * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vget(a0, j), vget(bfr, j), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vget(a1, j), vget(bfr, j), acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            a0 = n0;
            a1 = n1;
        }
        if (c + 1 < NCH) {
            store_chunk(c + 1);
            __syncthreads();
        }
    }
    __syncthreads();  // every wave is done reading the staged rows: the buffer becomes the gate exchange
    float *sG = smem;                  // [4 gates][32 rows][17]
    float *sHb = smem + 4 * MR * 17;   // split heads: relu(bn(h')) [32][17]
    float *sA2 = sHb + MR * 17;        //              the value | policy heads' input slice [32 rows][40] (36 used)
    float *sP = sA2 + MR * 40;         //              the partial block of this workgroup [32 rows][3 heads][32 hidden]
    if constexpr (SH) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = tid + NTHR * u;
            if (idx < 32 * 9) *reinterpret_cast<f32x4 *>(sA2 + (idx / 9) * 40 + (idx % 9) * 4) = sh_av[u];
        }
    }
    {
        const int col = lane & 15, rq = 4 * (lane >> 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sG[(wv * MR + rq + q) * 17 + col] = acc0[q];
            sG[(wv * MR + 16 + rq + q) * 17 + col] = acc1[q];
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int p = tid + NTHR * q, row = p >> 4, u = p & 15;
        const int b = r0 + row;
        if (b >= a.B) { if constexpr (SH) sHb[row * 17 + u] = 0.0f; continue; }
        const int unit = tile * 16 + u;
        const float gi = sG[(0 * MR + row) * 17 + u] + gb[q][0];
        const float gf = sG[(1 * MR + row) * 17 + u] + gb[q][1];
        const float gg = sG[(2 * MR + row) * 17 + u] + gb[q][2];
        const float go = sG[(3 * MR + row) * 17 + u] + gb[q][3];
        const float cn = sigmoidf_(gf) * c_prev[q] + sigmoidf_(gi) * tanhf_(gg);
        const float hn = sigmoidf_(go) * tanhf_(cn);
        const bool reset = a.search_len && a.horizon > 0 && (slen[q] % a.horizon) == 0;  // mcts_ctree.py:859-863
        store_wt(a.h_out + (size_t)b * H + unit, reset ? 0.0f : hn);
        store_wt(a.c_out + (size_t)b * H + unit, reset ? 0.0f : cn);
        const float hb = a.bn_scale ? fmaxf(hn * bns[q] + bnt[q], 0.0f) : hn;
        store_wt(a.hbn_out + (size_t)b * H + unit, hb);
        if constexpr (SH) sHb[row * 17 + u] = hb;
    }
    if constexpr (SH) {
        // value | policy heads, first layer: [32 rows x 36] x [36 x this wave's 16 of the 64 hidden columns], one MFMA chain per 16-row half
        {
            f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0;
#pragma unroll
            for (int ks = 0; ks < 9; ++ks) {
                const float bw = sh_bv[ks >> 2][ks & 3];
                p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(sA2[(lane & 15) * 40 + 4 * ks + (lane >> 4)], bw, p0, 0, 0, 0);
                p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(sA2[(16 + (lane & 15)) * 40 + 4 * ks + (lane >> 4)], bw, p1, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sP[(4 * (lane >> 4) + q) * 96 + (wv >> 1) * 32 + 16 * (wv & 1) + (lane & 15)] = p0[q];
                sP[(16 + 4 * (lane >> 4) + q) * 96 + (wv >> 1) * 32 + 16 * (wv & 1) + (lane & 15)] = p1[q];
            }
        }
        __syncthreads();   // relu(bn(h')) of all 32 x 16 (row, unit) pairs is in sHb
        {   // value-prefix head, first layer: wave (half = wv >> 1, column half = wv & 1): [16 rows x 16 units] x [16 x 16 hidden columns]
            const int half = wv >> 1;
            f32x4 pacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                pacc = __builtin_amdgcn_mfma_f32_16x16x4f32(sHb[(16 * half + (lane & 15)) * 17 + 4 * ks + (lane >> 4)], sh_brv[ks], pacc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) sP[(16 * half + 4 * (lane >> 4) + q) * 96 + 64 + 16 * (wv & 1) + (lane & 15)] = pacc[q];
        }
        __syncthreads();
        // the block goes out as whole 128-byte lines: (row, head) = 32 floats at [root][head][unit tile][32]
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = tid + NTHR * i;   // 32 rows x 24 float4
            const int row = idx / 24, c4 = idx % 24;
            if (r0 + row < a.B)
                *reinterpret_cast<f32x4 *>(a.sh_part + ((size_t)(r0 + row) * 3 + (c4 >> 3)) * (gridDim.x * 32) + (size_t)tile * 32 + (c4 & 7) * 4) =
                    *reinterpret_cast<const f32x4 *>(sP + row * 96 + c4 * 4);
        }
    }
    lz_stamp_end(a.stamp);
}

// ------------------------------------------------------------------------------------------------
// heads.  grid = (ceil(B/4), nheads), block = 256: four roots share every weight fetch.
// ------------------------------------------------------------------------------------------------
constexpr int EPB = 4;
constexpr int MAXH = 4;
struct head_pack { lz_head_desc h[MAXH]; unsigned long long *ts; };   // ts: debugging (s_memtime stamps of workgroup 0), null in production

// block-wide reduction of N values per thread (max or sum); result broadcast to every thread
template <int N, bool IS_MAX, int NWAVES>
__device__ __forceinline__ void block_reduce_n(float (&v)[N], float *scratch)
{
    // per-wave part on the DPP path (lz_wave.h): six ds_bpermute round trips per value otherwise sit on the kernel's
    // critical path (three dependent reductions per launch)
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = IS_MAX ? wave_max(v[k]) : wave_sum(v[k]);
    const int wv = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < N; ++k) scratch[wv * N + k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        float r[NWAVES / 4];
#pragma unroll
        for (int g = 0; g < NWAVES / 4; ++g) {
            const float a0 = scratch[(4 * g) * N + k], a1 = scratch[(4 * g + 1) * N + k], a2 = scratch[(4 * g + 2) * N + k], a3 = scratch[(4 * g + 3) * N + k];
            r[g] = IS_MAX ? fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)) : (a0 + a1) + (a2 + a3);
        }
        v[k] = r[0];
#pragma unroll
        for (int g = 1; g < NWAVES / 4; ++g) v[k] = IS_MAX ? fmaxf(v[k], r[g]) : v[k] + r[g];
    }
}

// NTHR threads: HID units x (NTHR / HID) K-parts in layer 1, ceil(768 / NTHR) outputs per thread in layer 2.  512 threads
// (8 waves) halve the per-thread load and FMA chains of the 256-thread version.
template <int HID, int NTHR>
__global__ __launch_bounds__(NTHR) void k_heads(head_pack hp, int B)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const lz_head_desc &h = hp.h[blockIdx.y];
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * EPB;
    const int K1 = h.K1;
    float *xs = smem;                    // [EPB][K1]
    float *hid = xs + EPB * K1;          // [EPB][HID]
    float *scr = hid + EPB * HID;        // [NWAVES * 8]
    constexpr int NWAVES = NTHR / 64, PARTS = NTHR / HID;  // PARTS lanes (a power of two) share one hidden unit's row
    // Every weight this thread will need is requested up front (none depends on the activations): the kernel is
    // a chain of three dependent stages, and each exposed L2 round trip costs as much as the arithmetic.
    constexpr int NW1 = 144 / PARTS, NPT = (768 + NTHR - 1) / NTHR;  // layer-1 float4 per thread kept in registers (K1 = 576
                                                                     // fits exactly); outputs per thread (NOUT <= 768)
    const int u = tid / PARTS, part = tid % PARTS;
    const float *wr = h.w1 + (size_t)min(u, HID - 1) * K1;
    constexpr int KSTEP = 4 * PARTS;
    const int n1 = (K1 - part * 4 + KSTEP - 1) / KSTEP;  // layer-1 iterations of this thread (k = part*4 + KSTEP i < K1)
    f32x4 w1r[NW1];
#pragma unroll
    for (int i = 0; i < NW1; ++i) w1r[i] = *reinterpret_cast<const f32x4 *>(wr + min(part * 4 + KSTEP * i, K1 - 4));
    float w2r[NPT][HID], b2r[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int n = min(tid + i * NTHR, h.NOUT - 1);
        b2r[i] = h.b2[n];
#pragma unroll
        for (int k4 = 0; k4 < HID / 4; ++k4) {
            const f32x4 w4 = *reinterpret_cast<const f32x4 *>(h.w2t + ((size_t)k4 * h.NOUT + n) * 4);
            w2r[i][4 * k4] = w4[0]; w2r[i][4 * k4 + 1] = w4[1]; w2r[i][4 * k4 + 2] = w4[2]; w2r[i][4 * k4 + 3] = w4[3];
        }
    }
    const float b1v = h.b1[min(u, HID - 1)], s1v = h.s1[min(u, HID - 1)], t1v = h.t1[min(u, HID - 1)];
    const bool stamp = hp.ts && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
#define LZ_HTS(i) do { if (stamp) hp.ts[i] = __builtin_readcyclecounter(); } while (0)
    LZ_HTS(0);
    for (int i = tid; i < EPB * (K1 / 4); i += NTHR) {  // 16-channel runs are contiguous: float4 loads
        const int e = i / (K1 / 4), k = (i - e * (K1 / 4)) * 4, b = min(b0 + e, B - 1);
        *reinterpret_cast<float4 *>(xs + e * K1 + k) =
            *reinterpret_cast<const float4 *>(h.in + (size_t)b * h.env_stride + (k >> 4) * h.pix_stride + (k & 15));
    }
    __syncthreads();
    LZ_HTS(1);
    // ---- layer 1: HID units x 8 K-parts; a unit's row is read 128 B at a time by its 8 lanes
    {
        float acc[EPB];
#pragma unroll
        for (int e = 0; e < EPB; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int i = 0; i < NW1; ++i) {
            if (i < n1) {
                const int k = part * 4 + KSTEP * i;
#pragma unroll
                for (int e = 0; e < EPB; ++e) {
                    const float4 xv = *reinterpret_cast<const float4 *>(xs + e * K1 + k);
                    acc[e] += w1r[i][0] * xv.x + w1r[i][1] * xv.y + w1r[i][2] * xv.z + w1r[i][3] * xv.w;
                }
            }
        }
        for (int i = NW1; i < n1; ++i) {  // K1 > 576 (board games): the tail streams from L2
            const int k = part * 4 + KSTEP * i;
            const f32x4 wv4 = *reinterpret_cast<const f32x4 *>(wr + k);
#pragma unroll
            for (int e = 0; e < EPB; ++e) {
                const float4 xv = *reinterpret_cast<const float4 *>(xs + e * K1 + k);
                acc[e] += wv4[0] * xv.x + wv4[1] * xv.y + wv4[2] * xv.z + wv4[3] * xv.w;
            }
        }
#pragma unroll
        for (int e = 0; e < EPB; ++e) acc[e] = group_sum<PARTS>(acc[e]);  // PARTS lanes of one DPP row share a hidden unit
        if (u < HID && part == 0) {
#pragma unroll
            for (int e = 0; e < EPB; ++e) hid[e * HID + u] = fmaxf((acc[e] + b1v) * s1v + t1v, 0.0f);
        }
    }
    __syncthreads();
    LZ_HTS(2);
    // ---- layer 2 on the transposed weights (already in registers)
    float lg[NPT][EPB];
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int n = tid + i * NTHR;
        const bool ok = n < h.NOUT;
        float acc[EPB];
#pragma unroll
        for (int e = 0; e < EPB; ++e) acc[e] = b2r[i];
#pragma unroll
        for (int k = 0; k < HID; ++k) {
#pragma unroll
            for (int e = 0; e < EPB; ++e) acc[e] += w2r[i][k] * hid[e * HID + k];
        }
#pragma unroll
        for (int e = 0; e < EPB; ++e) {
            lg[i][e] = ok ? acc[e] : -__builtin_inff();
            if (ok && h.out_logits && b0 + e < B) h.out_logits[(size_t)(b0 + e) * h.NOUT + n] = acc[e];
        }
    }
    LZ_HTS(3);
    if (!h.categorical) return;
    float m[EPB];
#pragma unroll
    for (int e = 0; e < EPB; ++e) {
        m[e] = lg[0][e];
#pragma unroll
        for (int i = 1; i < NPT; ++i) m[e] = fmaxf(m[e], lg[i][e]);
    }
    block_reduce_n<EPB, true, NWAVES>(m, scr);
    LZ_HTS(4);
    float ss[2 * EPB];
#pragma unroll
    for (int e = 0; e < EPB; ++e) {
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int n = tid + i * NTHR;
            if (n < h.NOUT) {
                const float ex = expf(lg[i][e] - m[e]);
                s0 += ex;
                s1 += ex * (h.support_min + (float)n);
            }
        }
        ss[2 * e] = s0;
        ss[2 * e + 1] = s1;
    }
    block_reduce_n<2 * EPB, false, NWAVES>(ss, scr);
    LZ_HTS(5);
    if (tid < EPB && b0 + tid < B) {
        // InverseScalarTransform.__call__ (scaling_transform.py:82-92), torch's fp32 op order
        float s0 = ss[0], s1 = ss[1];
#pragma unroll
        for (int e = 1; e < EPB; ++e) if (tid == e) { s0 = ss[2 * e]; s1 = ss[2 * e + 1]; }
        const float value = s1 / s0;
        h.out_scalar[b0 + tid] = lz_inverse_scalar_transform(value);
        if (h.out_expect) h.out_expect[b0 + tid] = value;   // parity tests (tracing): the pre-transform expectation
    }
    LZ_HTS(6);
#undef LZ_HTS
}

// ------------------------------------------------------------------------------------------------
// The same heads on the matrix pipe (K1 a multiple of 64 up to 1024, NOUT <= 640): four roots per workgroup are exactly the four rows
// of v_mfma_f32_4x4x1_f32, whose 16 blocks = (k quarter) x (quad of output units).  Phase stamps of the VALU kernel above showed
// where its 15 k cycles go: 3.8 k until the rows arrive (queued behind 151 KB of weights), 3.1 k in layer 1 (every thread re-reads
// its k slice of the four rows from LDS: 295 KB of LDS traffic for 9 KB of data), 3.4 k in layer 2, 4.2 k in the softmax's block
// reductions.  Here: rows are requested first; layer 1 = 36 MFMAs per wave (waves = 4 k quarters x 2 halves of the hidden units),
// 9 LDS reads per wave; layer 2 = 8 MFMAs per group of 16 outputs (waves take groups w, w + 8, ...), weights straight from L2 in the
// [HID/4][NOUT][4] layout the VALU kernel uses; after the k-quarter partial sums are added across lanes every quarter holds all four
// roots, so quarter q keeps root q: max / exp / sums then run on 16-lane DPP rows and 8 x 4 values cross the waves.
// grid = (ceil(B / 4), nheads), block = 512.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float row16_max(float v)   // max over the 16 lanes of a DPP row, in every lane of the row
{
    int x = __float_as_int(v);
#define LZ_DPPR(ctrl) __int_as_float(__builtin_amdgcn_update_dpp(x, x, ctrl, 0xf, 0xf, false))
    v = fmaxf(v, LZ_DPPR(0xB1)); x = __float_as_int(v);
    v = fmaxf(v, LZ_DPPR(0x4E)); x = __float_as_int(v);
    v = fmaxf(v, LZ_DPPR(0x141)); x = __float_as_int(v);
    v = fmaxf(v, LZ_DPPR(0x140));
#undef LZ_DPPR
    return v;
}

template <int MAXG>   // K1 <= 64 MAXG: 9 (6x6 latent: 16 x 36 head inputs, LSTM 512) | 16 (8x8 latent: 16 x 64) | 21 (9x9 board: 16 x 81)
__global__ __launch_bounds__(512) void k_heads_mm(head_pack hp, int B)
{
    constexpr int HID = 32, MAXT = 5;   // NOUT <= 128 MAXT
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const lz_head_desc &h = hp.h[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n16 = lane & 15, kq = lane >> 4;
    const int b0 = blockIdx.x * EPB, K1 = h.K1, NOUT = h.NOUT, NG1 = ((K1 >> 4) + 3) >> 2;   // 16-wide k groups per k-quarter wave (the last quarter may hold fewer:
                                                                                            // K1 = 16 x 81 on a 9x9 board = 21 + 21 + 21 + 18 groups)
    float *xs = smem;                    // [4][K1]
    float *part = xs + EPB * K1;         // [8 waves][4 roots][16 units]
    float *hid = part + 8 * 64;          // [4][32]
    float *scrm = hid + EPB * HID;       // [8 waves][4 roots]
    float *scrs = scrm + 32;             // [8 waves][4 roots][2]
    const bool stamp = hp.ts && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
#define LZ_HTS(i) do { if (stamp) hp.ts[i] = __builtin_readcyclecounter(); } while (0)
    // ---- requests, in the order of use (a wave's loads return in order): the four rows, layer-1 fragments, layer-2 fragments
    constexpr int NXV = (EPB * 64 * MAXG / 4 + 511) / 512;
    f32x4 xv[NXV];   // (native vectors: an array of HIP's float4 struct lands in scratch memory here)
#pragma unroll
    for (int u = 0; u < NXV; ++u) {
        const int i = min(tid + 512 * u, EPB * (K1 / 4) - 1);
        const int e = i / (K1 / 4), k = (i - e * (K1 / 4)) * 4, b = min(b0 + e, B - 1);
        xv[u] = *reinterpret_cast<const f32x4 *>(h.in + (size_t)b * h.env_stride + (k >> 4) * h.pix_stride + (k & 15));
    }
    const int kw = wv & 3, uh = wv >> 2;   // layer 1: this wave's k quarter and half of the hidden units
    const int ng = max(0, min(NG1, (K1 >> 4) - kw * NG1));   // groups of this wave's quarter
    f32x4 w1f[MAXG];
    {
        const float *wr = h.w1 + (size_t)(16 * uh + n16) * K1 + (size_t)(ng > 0 ? kw * NG1 * 16 : 0) + kq * 4;
#pragma unroll
        for (int g = 0; g < MAXG; ++g) w1f[g] = *reinterpret_cast<const f32x4 *>(wr + min(g, max(ng - 1, 0)) * 16);
    }
    f32x4 w2f[MAXT][2];
    float b2v[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const int n = min(16 * (wv + 8 * t) + n16, NOUT - 1);
        b2v[t] = h.b2[n];
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) w2f[t][sp] = *reinterpret_cast<const f32x4 *>(h.w2t + ((size_t)(4 * sp + kq) * NOUT + n) * 4);
    }
    const int hu = tid & 31;
    const float b1v = h.b1[hu], s1v = h.s1[hu], t1v = h.t1[hu];
    LZ_HTS(0);
#pragma unroll
    for (int u = 0; u < NXV; ++u) {
        const int i = tid + 512 * u;
        if (i < EPB * (K1 / 4)) *reinterpret_cast<f32x4 *>(xs + (size_t)i * 4) = xv[u];
    }
    __syncthreads();
    LZ_HTS(1);
    // ---- layer 1: D[root][unit] partial over this wave's k quarter (and, inside the instruction, over the four k of a step)
    {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float *xa = xs + (lane & 3) * K1 + kw * NG1 * 16 + kq * 4;
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            if (g < ng) {   // wave-uniform
                const f32x4 a = *reinterpret_cast<const f32x4 *>(xa + g * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j], w1f[g][j], acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = xor32_sum(xor16_sum(acc[i]));
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) part[(wv * 4 + i) * 16 + n16] = acc[i];
        }
    }
    __syncthreads();
    if (tid < EPB * HID) {   // (root, unit): the four k quarters in order, then bias, BatchNorm, ReLU
        const int r = tid >> 5, u = tid & 31, w0 = (u >> 4) * 4, n = u & 15;
        float sum = part[((w0 + 0) * 4 + r) * 16 + n];
        sum += part[((w0 + 1) * 4 + r) * 16 + n];
        sum += part[((w0 + 2) * 4 + r) * 16 + n];
        sum += part[((w0 + 3) * 4 + r) * 16 + n];
        hid[r * HID + u] = fmaxf((sum + b1v) * s1v + t1v, 0.0f);
    }
    __syncthreads();
    LZ_HTS(2);
    // ---- layer 2: output groups wv, wv + 8, ...; after the reduction lane (quarter q, n) keeps root q
    float lg[MAXT];
    {
        f32x4 a2[2];
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) a2[sp] = *reinterpret_cast<const f32x4 *>(hid + (lane & 3) * HID + 16 * sp + kq * 4);
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            lg[t] = -__builtin_inff();
            if (16 * (wv + 8 * t) < NOUT) {   // wave-uniform
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a2[sp][j], w2f[t][sp][j], acc, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = xor32_sum(xor16_sum(acc[i]));
                const float mine = kq == 0 ? acc[0] : kq == 1 ? acc[1] : kq == 2 ? acc[2] : acc[3];
                const int n = 16 * (wv + 8 * t) + n16;
                if (n < NOUT) {
                    lg[t] = mine + b2v[t];
                    if (h.out_logits && b0 + kq < B) h.out_logits[(size_t)(b0 + kq) * NOUT + n] = lg[t];
                }
            }
        }
    }
    LZ_HTS(3);
    if (!h.categorical) return;
    // ---- softmax . support of root kq on the 16-lane row, then across the 8 waves
    float m = lg[0];
#pragma unroll
    for (int t = 1; t < MAXT; ++t) m = fmaxf(m, lg[t]);
    m = row16_max(m);
    if (n16 == 0) scrm[wv * 4 + kq] = m;
    __syncthreads();
    m = scrm[kq];
#pragma unroll
    for (int w = 1; w < 8; ++w) m = fmaxf(m, scrm[w * 4 + kq]);
    LZ_HTS(4);
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const int n = 16 * (wv + 8 * t) + n16;
        if (n < NOUT) {
            const float ex = expf(lg[t] - m);
            s0 += ex;
            s1 += ex * (h.support_min + (float)n);
        }
    }
    s0 = group_sum<16>(s0);
    s1 = group_sum<16>(s1);
    if (n16 == 0) { scrs[(wv * 4 + kq) * 2] = s0; scrs[(wv * 4 + kq) * 2 + 1] = s1; }
    __syncthreads();
    LZ_HTS(5);
    if (tid < EPB && b0 + tid < B) {
        float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; ++w) { t0 += scrs[(w * 4 + tid) * 2]; t1 += scrs[(w * 4 + tid) * 2 + 1]; }
        // InverseScalarTransform.__call__ (scaling_transform.py:82-92), torch's fp32 op order
        const float value = t1 / t0;
        h.out_scalar[b0 + tid] = lz_inverse_scalar_transform(value);
        if (h.out_expect) h.out_expect[b0 + tid] = value;   // parity tests (tracing): the pre-transform expectation
    }
    LZ_HTS(6);
#undef LZ_HTS
}

__global__ __launch_bounds__(256) void k_hinv_nn(const float *__restrict__ in, float *__restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = lz_inverse_scalar_transform(in[i]);
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

template <int CIN, int COUT, int STRIDE, int TM>
static void launch_big(const lz_conv_args &a, hipStream_t s)
{
    const int M = a.B * a.Hout * a.Wout;
    int npix = (STRIDE == 1) ? TM + 2 * a.Win + 2 : 2 * a.Win * ((TM - 1) / a.Wout + 1) + 2 * (a.Wout - 1) + 2 * a.Win + 3;
    npix = min(npix, a.B * a.Hin * a.Win);
    const size_t lds = (size_t)(npix + 1) * (CIN + 4) * 4;
    hipLaunchKernelGGL((k_conv3x3_big<CIN, COUT, STRIDE, TM>), dim3((M + TM - 1) / TM), dim3(256), lds, s, a, npix);
}

template <int CIN, int COUT, int TMT>
static void launch_wino(const lz_conv_args &a, hipStream_t s)
{
    const int ntiles = a.B * (a.Hout / 2) * (a.Wout / 2);
    const size_t lds = (size_t)8 * TMT * (CIN + 4) * 4;
    hipLaunchKernelGGL((k_conv_wino<CIN, COUT, TMT>), dim3((ntiles + TMT - 1) / TMT), dim3(256), lds, s, a);
}

void lz_launch_hinv_nn(const float *d_in, float *d_out, int64_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_hinv_nn, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_in, d_out, n);
}

// the downsample block's two stride-2 convolutions of one input in one launch (k_conv_s3<32, 64, 2, true>); false: not applicable --
// the caller launches them one after the other.  LZ_CONV_NO_DUAL=1 keeps the two launches (A/B runs).
bool lz_launch_conv3x3_pair(const lz_conv_args &a, const lz_conv_args &b, int cin, int stride, hipStream_t s)
{
    static const char *off = getenv("LZ_CONV_NO_DUAL"), *nosplit = getenv("LZ_CONV_NO_SPLIT"), *direct = getenv("LZ_CONV_DIRECT");
    if (off || nosplit || direct) return false;
    if (!(cin == 32 && a.Cout == 64 && b.Cout == 64 && stride == 2) || !a.w3 || !b.w3 || a.act_bf16 || b.act_bf16) return false;
    if (a.in != b.in || a.gather_ix || b.gather_ix || a.act_table || b.act_table || a.residual || b.residual) return false;
    if (a.B != b.B || a.Hin != b.Hin || a.Win != b.Win || a.Hout != b.Hout || a.Wout != b.Wout || 96 % a.Wout != 0) return false;
    const int TR = 96 / a.Wout;
    const int HR = (TR - 1) * 2 + 3, HC = (a.Wout - 1) * 2 + 3;
    constexpr int PB = 5 * 8, NLD = 7;
    if (HR * HC * (32 / 4) > NLD * 512) return false;
    const size_t lds = (size_t)3 * (((size_t)HR * HC * PB + 7) & ~(size_t)7) * 2 + 2 * 2 * 64 * 4;
    if (lds > 150 * 1024) return false;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void *)k_conv_s3<32, 64, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr_set = true; }
    const int ntiles = a.B * ((a.Hout + TR - 1) / TR);
    const int grid = ntiles < 256 ? ntiles : 256;   // persistent, one workgroup per CU
    lz_conv_second d{b.w3, b.scale, b.shift, b.out, b.relu};
    hipLaunchKernelGGL((k_conv_s3<32, 64, 2, true>), dim3(grid), dim3(512), lds, s, a, ntiles, TR, d);
    return true;
}

void lz_launch_conv3x3(const lz_conv_args &a, int cin, int stride, hipStream_t s)
{
    // fast mode (lz_model_cfg::precision = 1): the layer carries bf16 fragments
    if (a.wb && a.act_bf16 && !a.gather_ix && !a.act_table && (a.Wout == 48 || a.Wout == 24 || a.Wout == 12 || a.Wout == 32 || a.Wout == 16 || a.Wout == 8)) {   // (the halo of these grids -- 96x96 and 64x64 observations -- fits the kernel's per-thread piece count)
        if (cin == 32 && a.Cout == 32 && stride == 1) { launch_conv_bf<32, 32, 1>(a, s); return; }
        if (cin == 32 && a.Cout == 64 && stride == 2) { launch_conv_bf<32, 64, 2>(a, s); return; }
        if (cin == 64 && a.Cout == 64 && stride == 1) { launch_conv_bf<64, 64, 1>(a, s); return; }
    }
    // parity mode, round 5: the tower's convolutions as split-bf16 products (k_conv_s3: fp32 accuracy on the bf16 matrix pipe); LZ_CONV_NO_SPLIT=1
    // keeps the fp32-matrix kernels below (Winograd / direct) -- the A/B switch of tests/test_kernel_variants_gpu.py
    static const char *nosplit = getenv("LZ_CONV_NO_SPLIT");
    static const char *direct = getenv("LZ_CONV_DIRECT");
    if (a.w3 && !a.act_bf16 && !nosplit && !direct && !a.gather_ix && !a.act_table && !a.residual_gather) {
        if (cin == 32 && a.Cout == 32 && stride == 1 && launch_conv_s3<32, 32, 1>(a, s)) return;
        if (cin == 32 && a.Cout == 64 && stride == 2 && launch_conv_s3<32, 64, 2>(a, s)) return;
        if (cin == 64 && a.Cout == 64 && stride == 1 && launch_conv_s3<64, 64, 1>(a, s)) return;
    }
    // stride-1 convolutions of the tower: Winograd F(2x2, 3x3) when the transformed weights exist (LZ_CONV_DIRECT=1: the direct form)
    if (a.uf && !direct && stride == 1 && !a.gather_ix && !a.act_table && a.Hout >= 12 && (a.Hout & 1) == 0 && (a.Wout & 1) == 0) {
        // tiles per workgroup: 32 for the 32-channel layers (one (tile, channel-quad) item per thread); 16 for the 64-channel layers
        // (16 accumulator tiles of 4 registers + one item per thread = 148 registers, three workgroups per CU: 48 us per layer on
        // average against 73 us with 32 tiles / one workgroup per CU and 79 us for the direct kernel)
        if (cin == 32 && a.Cout == 32) { launch_wino<32, 32, 32>(a, s); return; }
        if (cin == 64 && a.Cout == 64) { launch_wino<64, 64, 16>(a, s); return; }
    }
    if (a.wf && !a.gather_ix && !a.act_table && a.Hout >= 12) {  // representation tower: big-grid kernel
        if (cin == 32 && a.Cout == 32 && stride == 1) { launch_big<32, 32, 1, 128>(a, s); return; }
        if (cin == 32 && a.Cout == 64 && stride == 2) { launch_big<32, 64, 2, 48>(a, s); return; }
        if (cin == 64 && a.Cout == 64 && stride == 1) { launch_big<64, 64, 1, 144>(a, s); return; }
    }
    const int M = a.B * a.Hout * a.Wout;
    int npix = conv_npix_max(a, stride);
    npix = min(npix, a.B * a.Hin * a.Win);
    const int ps = cin + 4;
    size_t lds = ((size_t)(npix + 1) * ps + (size_t)9 * 16 * ps) * 4;
    const size_t red = (size_t)4 * 9 * 4 * 64 * 4;
    if (lds < red) lds = red;
    dim3 grid((M + 143) / 144, a.Cout / 16), block(256);
    if (cin == 64 && stride == 1) hipLaunchKernelGGL((k_conv3x3<64, 1>), grid, block, lds, s, a, npix);
    else if (cin == 32 && stride == 1) hipLaunchKernelGGL((k_conv3x3<32, 1>), grid, block, lds, s, a, npix);
    else if (cin == 32 && stride == 2) hipLaunchKernelGGL((k_conv3x3<32, 2>), grid, block, lds, s, a, npix);
    else if (cin == 64 && stride == 2) hipLaunchKernelGGL((k_conv3x3<64, 2>), grid, block, lds, s, a, npix);
}

void lz_launch_conv_first(const float *obs, const float *w, const float *scale, const float *shift, float *out, int B,
                          int C, int H, int W, int Cout, hipStream_t s, int out_bf16)
{
    const int64_t M = (int64_t)B * (H / 2) * (W / 2);
    dim3 grid((unsigned)((M + 63) / 64)), block(256);
    static const char *valu = getenv("LZ_CONV_FIRST_VALU");   // A/B switch: the VALU kernel (another summation order, same tolerance)
    if (C == 4 && Cout == 32 && !valu && (W == 96 || W == 64) && H == W) {   // the Atari shapes: rows through LDS, taps on the matrix pipe
        const int TR = 96 / (W / 2), bands = (H / 2 + TR - 1) / TR;
        const size_t lds = (size_t)4 * (2 * TR + 1) * (W + 8) * 4;
        if (out_bf16) hipLaunchKernelGGL((k_conv_first_mm<true>), dim3(B * bands), block, lds, s, obs, w, scale, shift, out, B, H, W, TR);
        else hipLaunchKernelGGL((k_conv_first_mm<false>), dim3(B * bands), block, lds, s, obs, w, scale, shift, out, B, H, W, TR);
        return;
    }
    if (out_bf16) {   // fast mode: 4 x 96 x 96 observations only (lz_model_create)
        hipLaunchKernelGGL((k_conv_first<4, 32, true>), grid, block, 0, s, obs, w, scale, shift, out, B, H, W);
        return;
    }
    if (C == 4 && Cout == 32) hipLaunchKernelGGL((k_conv_first<4, 32>), grid, block, 0, s, obs, w, scale, shift, out, B, H, W);
    else if (C == 1 && Cout == 32) hipLaunchKernelGGL((k_conv_first<1, 32>), grid, block, 0, s, obs, w, scale, shift, out, B, H, W);
    else if (C == 3 && Cout == 32) hipLaunchKernelGGL((k_conv_first<3, 32>), grid, block, 0, s, obs, w, scale, shift, out, B, H, W);
    else if (C == 12 && Cout == 32) hipLaunchKernelGGL((k_conv_first<12, 32>), grid, block, 0, s, obs, w, scale, shift, out, B, H, W);
}

void lz_launch_conv_in(const float *obs, const float *w, const float *scale, const float *shift, float *out, int B, int C,
                       int H, int W, int Cout, hipStream_t s)
{
    const int64_t M = (int64_t)B * H * W;
    const int ppb = 256 / (Cout / 8);
    const dim3 grid((unsigned)((M + ppb - 1) / ppb));
    const size_t lds = (size_t)9 * C * Cout * 4;
    if (Cout == 64) hipLaunchKernelGGL(k_conv_in<64>, grid, dim3(256), lds, s, obs, w, scale, shift, out, B, C, H, W);
    else if (Cout == 32) hipLaunchKernelGGL(k_conv_in<32>, grid, dim3(256), lds, s, obs, w, scale, shift, out, B, C, H, W);
    else if (Cout == 16) hipLaunchKernelGGL(k_conv_in<16>, grid, dim3(256), lds, s, obs, w, scale, shift, out, B, C, H, W);
}

void lz_launch_avgpool(const float *in, float *out, int B, int Hin, int Win, int C, hipStream_t s, int in_bf16, int out_bf16)
{
    const int Ho = (Hin + 1) / 2, Wo = (Win + 1) / 2;
    if (in_bf16) {
        const int64_t n8 = (int64_t)B * Ho * Wo * (C / 8);
        const dim3 g((unsigned)((n8 + 255) / 256));
        if (out_bf16) hipLaunchKernelGGL((k_avgpool_bf<true>), g, dim3(256), 0, s, reinterpret_cast<const __bf16 *>(in), (void *)out, B, Hin, Win, C);
        else hipLaunchKernelGGL((k_avgpool_bf<false>), g, dim3(256), 0, s, reinterpret_cast<const __bf16 *>(in), (void *)out, B, Hin, Win, C);
        return;
    }
    const int64_t n = (int64_t)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(k_avgpool, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, B, Hin, Win, C);
}

void lz_launch_bf16_to_f32(const void *in, float *out, size_t n, hipStream_t s)
{
    const size_t n8 = n / 8;
    hipLaunchKernelGGL(k_bf16_to_f32, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const __bf16 *>(in), out, n8);
}

void lz_launch_conv1x1(const lz_c1_args &a, hipStream_t s)
{
    hipLaunchKernelGGL(k_conv1x1, dim3((a.npix + 143) / 144, a.njobs), dim3(256), 0, s, a);
}

// the tree step can ride in the chain's prologue when the root's tree fits the (still unused) activation buffers, one lane
// per action covers the node, and a gather / action table is what the chain would have read anyway
bool lz_chain_fusable(const lz_chain_args &a, const lz_tree_step &step)
{
    if (a.C != 0 && a.C != 64) return false;  // the narrow chain has no fused instance
    if (a.gelu) return false;                 // GELU networks have the plain chain instance only
    if (!((a.gw == 6 && a.gh == 6) || (a.gw == 8 && a.gh == 8)) || a.tstamp || !a.gather_ix || !a.act_table) return false;
    if (step.t.A > 64 || step.t.B != a.B) return false;
    if (step.t.variant != LZ_TREE_EFFICIENTZERO && step.t.variant != LZ_TREE_MUZERO) return false;
    // the staged tree lives in the four activation buffers (4 x 37 x 68 floats = 40 KB) until the latent is loaded over it;
    // staging more than 16 KB per root and simulation was measured slower than walking the HBM arrays (configs[2], 400 sims)
    const size_t room = (size_t)4 * (a.gw * a.gh + 1) * 68 * 4, lim = lz_tree_lds_limit(16 * 1024);
    // MuZero batches of more than two roots per CU (BASELINE configs[2]: 1024 roots): the prologue's one-wave step is serial time on a
    // CU that queues four workgroups, while the separate workgroup-parallel step (k_tree_step_wg: A <= 8) runs every root at once --
    // measured 68.9 -> 67.3 ms per 1024 x 400 step with the step kept separate for all simulations.
    // (EfficientZero keeps the fused form at every batch size: its split heads ride on the same prologue.)
    if (step.t.variant == LZ_TREE_MUZERO && step.t.B > 512 && step.t.A <= 8 && !step.a.serial && !getenv("LZ_TREE_LDS_LIMIT") && !getenv("LZ_TREE_NO_WG")) return false;
    return lz_tree_lds_bytes(step.t, step.new_node) <= (lim < room ? lim : room);
}

template <int GW, int GH, int C>
static void launch_chain_small(const lz_chain_args &a, hipStream_t s)
{
    constexpr int HW = GW * GH, PS = C + 4;
    const size_t lds = (size_t)(4 * (HW + 1) * PS + HW * PS + LZ_CHAIN_MAX_LAYERS * 2 * C) * 4;
    hipLaunchKernelGGL((k_chain_small<GW, GH, C>), dim3(a.B), dim3(256), lds, s, a);
}

// grids / widths with a narrow-chain instance (board games: tictactoe 3x3, gomoku 6x6, connect4 6x7, Go 9x9)
bool lz_chain_small_supported(int gw, int gh, int C)
{
    if (C != 16 && C != 32) return false;
    return (gw == 3 && gh == 3) || (gw == 6 && gh == 6) || (gw == 7 && gh == 6) || (gw == 9 && gh == 9);
}

void lz_launch_chain(const lz_chain_args &a, hipStream_t s, const lz_tree_step *step)
{
    if (a.C == 16 || a.C == 32) {  // narrow networks: k_chain_small, never tree-fused
#define LZ_SMALL(GWv, GHv) if (a.gw == GWv && a.gh == GHv) { if (a.C == 32) launch_chain_small<GWv, GHv, 32>(a, s); else launch_chain_small<GWv, GHv, 16>(a, s); return; }
        LZ_SMALL(3, 3) LZ_SMALL(6, 6) LZ_SMALL(7, 6) LZ_SMALL(9, 9)
#undef LZ_SMALL
        return;
    }
    auto lds_of = [](int hw, int extra) { return (size_t)(4 * (hw + 1) * 68 + hw * 68 + LZ_CHAIN_MAX_LAYERS * 128 + extra) * 4; };
    // fast mode (lz_model_cfg::precision = 1): every layer carries bf16 fragments -> k_chain_b
    {
        bool fast = ((a.gw == 6 && a.gh == 6) || (a.gw == 8 && a.gh == 8)) && a.nlayers > 0 && !a.tstamp;
        for (int i = 0; i < a.nlayers; ++i) fast = fast && a.layer[i].wb != nullptr;
        if (fast) {
            const int hw = a.gw * a.gh, mt = (hw + 15) / 16;
            const size_t lds = (size_t)(4 * (hw + 1) * 68 + hw * 68 + LZ_CHAIN_MAX_LAYERS * 128 + 128 + 4 * mt * 256) * 4 + (size_t)4 * (hw + 1) * 80 * 2;
            const dim3 g(a.B), blk(512);
            if (a.gw == 8) {   // 8x8 latent (64x64 observations): no split heads
                if (step && step->t.variant == LZ_TREE_EFFICIENTZERO) hipLaunchKernelGGL((k_chain_b<8, 8, 1, false>), g, blk, lds, s, a, *step);
                else if (step) hipLaunchKernelGGL((k_chain_b<8, 8, 2, false>), g, blk, lds, s, a, *step);
                else hipLaunchKernelGGL((k_chain_b<8, 8>), g, blk, lds, s, a, no_step{});
                return;
            }
            if (step) {
                if (step->t.variant == LZ_TREE_EFFICIENTZERO && step->sh.on) hipLaunchKernelGGL((k_chain_b<6, 6, 1, true>), g, blk, lds, s, a, *step);
                else if (step->t.variant == LZ_TREE_EFFICIENTZERO) hipLaunchKernelGGL((k_chain_b<6, 6, 1, false>), g, blk, lds, s, a, *step);
                else hipLaunchKernelGGL((k_chain_b<6, 6, 2, false>), g, blk, lds, s, a, *step);
            } else {
                hipLaunchKernelGGL((k_chain_b<6, 6>), g, blk, lds, s, a, no_step{});
            }
            return;
        }
    }
    // parity mode, 6x6 grid, every layer with split-bf16 planes: k_chain_s3 (LZ_CHAIN_NO_SPLIT=1 keeps the fp32-matrix chains below)
    static const char *direct = getenv("LZ_CHAIN_DIRECT");
    {
        static const char *nosplit = getenv("LZ_CHAIN_NO_SPLIT");
        // the other grids (8x8, 9x9, 6x7, 4x4) and the GELU networks: k_chain_s3g (lz_chain_s3g.hip)
        if (!nosplit && !direct && !getenv("LZ_CHAIN_W4") && lz_launch_chain_s3g(a, s, step)) return;
        bool s3 = !nosplit && !direct && !getenv("LZ_CHAIN_W4") && a.gw == 6 && a.gh == 6 && a.nlayers > 0 && !a.tstamp && !a.gelu && (a.C == 0 || a.C == 64);
        for (int i = 0; i < a.nlayers; ++i) s3 = s3 && a.layer[i].w3 != nullptr;
        if (s3) {
            const size_t lds = LZ_S3_LDS;
            const dim3 g(a.B), blk(512);
            static bool attr = false;
            if (!attr) {
                (void)hipFuncSetAttribute((const void *)k_chain_s3<6, 6, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void *)k_chain_s3<6, 6, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void *)k_chain_s3<6, 6, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void *)k_chain_s3<6, 6, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attr = true;
            }
            if (step) {
                if (step->t.variant == LZ_TREE_EFFICIENTZERO && step->sh.on) hipLaunchKernelGGL((k_chain_s3<6, 6, 1, true>), g, blk, lds, s, a, *step);
                else if (step->t.variant == LZ_TREE_EFFICIENTZERO) hipLaunchKernelGGL((k_chain_s3<6, 6, 1, false>), g, blk, lds, s, a, *step);
                else hipLaunchKernelGGL((k_chain_s3<6, 6, 2, false>), g, blk, lds, s, a, *step);
            } else {
                hipLaunchKernelGGL((k_chain_s3<6, 6>), g, blk, lds, s, a, no_step{});
            }
            return;
        }
    }
    // 6x6 / 8x8 grids whose layers all carry Winograd-transformed weights: k_chain_w (LZ_CHAIN_DIRECT=1: the direct form)
    bool wino = (!direct || a.gelu) && ((a.gw == 6 && a.gh == 6) || (a.gw == 8 && a.gh == 8)) && a.nlayers > 0;   // (only k_chain_w reads the GELU codes)
    for (int i = 0; i < a.nlayers; ++i) wino = wino && a.layer[i].uc != nullptr;
    if (wino) {
        static const char *w4 = getenv("LZ_CHAIN_W4");  // 4 waves (one per SIMD) instead of 8
        const int hw = a.gw * a.gh, nt = hw / 4, nw = (w4 && a.gw == 6 && !a.gelu) ? 4 : 8;   // (8x8 has no 4-wave instance)
        const bool big = hw > 36;
        const size_t lds = (size_t)(4 * (hw + 1) * 68 + (big ? 0 : hw * 68) + LZ_CHAIN_MAX_LAYERS * 128 + 128 + 16 * nt * 68 +
                                    (big ? 0 : nw * nt * 2 * 64)) * 4;
        const dim3 g(a.B), blk(nw * 64);
        if (a.gelu) {   // GELU networks (conv Sampled EfficientZero): never tree-fused (lz_chain_fusable), 8 waves
            if (a.gw == 6) hipLaunchKernelGGL((k_chain_w<6, 6, 8, false, 0, 16, false, true>), g, dim3(512), lds, s, a, no_step{});
            else hipLaunchKernelGGL((k_chain_w<8, 8, 8, false, 0, 8, false, true>), g, dim3(512), lds, s, a, no_step{});
            return;
        }
#define LZ_W(GWv, NWv) \
        if (step) { \
            if (step->t.variant == LZ_TREE_EFFICIENTZERO && step->sh.on && GWv == 6 && NWv == 8) hipLaunchKernelGGL((k_chain_w<6, 6, 8, false, 1, 16, true>), g, blk, lds, s, a, *step); \
            else if (step->t.variant == LZ_TREE_EFFICIENTZERO) hipLaunchKernelGGL((k_chain_w<GWv, GWv, NWv, false, 1>), g, blk, lds, s, a, *step); \
            else hipLaunchKernelGGL((k_chain_w<GWv, GWv, NWv, false, 2>), g, blk, lds, s, a, *step); \
        } else if (a.tstamp) { \
            hipLaunchKernelGGL((k_chain_w<GWv, GWv, NWv, true>), g, blk, lds, s, a, no_step{}); \
        } else { \
            hipLaunchKernelGGL((k_chain_w<GWv, GWv, NWv>), g, blk, lds, s, a, no_step{}); \
        }
        if (a.gw == 6) { if (nw == 4) { LZ_W(6, 4) } else { LZ_W(6, 8) } }
        else { LZ_W(8, 8) }   // (8x8 has no 4-wave instance: its 512 transform items are one per thread of 8 waves)
#undef LZ_W
        return;
    }
    if (step) {
        const bool ez = step->t.variant == LZ_TREE_EFFICIENTZERO;
        if (a.gw == 8) {
            if (ez) hipLaunchKernelGGL((k_chain<8, 8, false, 1>), dim3(a.B), dim3(256), lds_of(64, 4), s, a, *step);
            else hipLaunchKernelGGL((k_chain<8, 8, false, 2>), dim3(a.B), dim3(256), lds_of(64, 4), s, a, *step);
        } else {
            if (ez) hipLaunchKernelGGL((k_chain<6, 6, false, 1>), dim3(a.B), dim3(256), lds_of(36, 4), s, a, *step);
            else hipLaunchKernelGGL((k_chain<6, 6, false, 2>), dim3(a.B), dim3(256), lds_of(36, 4), s, a, *step);
        }
        return;
    }
    if (a.gw == 8 && a.gh == 8) { hipLaunchKernelGGL((k_chain<8, 8>), dim3(a.B), dim3(256), lds_of(64, 0), s, a, no_step{}); return; }
    if (a.gw == 7 && a.gh == 6) { hipLaunchKernelGGL((k_chain<7, 6>), dim3(a.B), dim3(256), lds_of(42, 0), s, a, no_step{}); return; }
    if (a.gw == 4 && a.gh == 4) { hipLaunchKernelGGL((k_chain<4, 4>), dim3(a.B), dim3(256), lds_of(16, 0), s, a, no_step{}); return; }   // 2048
    if (a.gw == 6 && a.gh == 6 && a.tstamp) hipLaunchKernelGGL((k_chain<6, 6, true>), dim3(a.B), dim3(256), lds_of(36, 64), s, a, no_step{});
    else if (a.gw == 6 && a.gh == 6) hipLaunchKernelGGL((k_chain<6, 6>), dim3(a.B), dim3(256), lds_of(36, 0), s, a, no_step{});
    else if (a.gw == 9 && a.gh == 9) hipLaunchKernelGGL((k_chain<9, 9>), dim3(a.B), dim3(256), lds_of(81, 0), s, a, no_step{});
}

template <int MROWS>
static void launch_lstm_m(const lz_lstm_args &a, hipStream_t s)
{
    // LDS: double-buffered A [MROWS][68] + B [32][68] chunks; reused for the split-K reduction + gate staging
    size_t lds = (size_t)(2 * MROWS * 68 + 2 * 32 * 68) * 4;
    const size_t red = (size_t)(4 * (MROWS / 16) * 2 * 4 * 64 + MROWS * 33) * 4;
    if (lds < red) lds = red;
    dim3 grid((4 * a.H) / 32, (a.B + MROWS - 1) / MROWS), block(256);
    const int nchunk = (a.KX + a.H) / 64;
    if (nchunk == 17) hipLaunchKernelGGL((k_lstm<17, MROWS>), grid, block, lds, s, a);
    else if (nchunk == 13) hipLaunchKernelGGL((k_lstm<13, MROWS>), grid, block, lds, s, a);
    else if (nchunk == 9) hipLaunchKernelGGL((k_lstm<9, MROWS>), grid, block, lds, s, a);
    else if (nchunk == 12) hipLaunchKernelGGL((k_lstm<12, MROWS>), grid, block, lds, s, a);  // MLP models: latent 256 + hidden 512
    else if (nchunk == 4) hipLaunchKernelGGL((k_lstm<4, MROWS>), grid, block, lds, s, a);    // latent 128 + hidden 128
    else if (nchunk == 8) hipLaunchKernelGGL((k_lstm<8, MROWS>), grid, block, lds, s, a);    // latent 256 + hidden 256
    else if (nchunk == 6) hipLaunchKernelGGL((k_lstm<6, MROWS>), grid, block, lds, s, a);    // 128 + 256 | 256 + 128
}

void lz_lstm_pack_fragments(const float *wcat, int H, int K, float *out)
{
    const int NKB = K / 16;
    for (int t = 0; t < H / 16; ++t)
        for (int g = 0; g < 4; ++g)
            for (int kb = 0; kb < NKB; ++kb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        const int n = lane & 15, kq = lane >> 4;
                        out[((((size_t)(t * 4 + g) * NKB + kb) * 64) + lane) * 4 + j] =
                            wcat[(size_t)(4 * (16 * t + n) + g) * K + 16 * kb + 4 * kq + j];
                    }
}

static bool launch_lstm2(const lz_lstm_args &a, hipStream_t s)
{
    static const char *off = getenv("LZ_LSTM_CHUNKED");
    if (a.wb) {   // fast mode (lz_model_finalize builds the bf16 fragments for these shapes only): 576 | 1024 + 512 columns, no input transform
        const dim3 g(a.H / 16, (a.B + 15) / 16);
        const int nkb = (a.KX + a.H) / 16;
        const size_t lds = std::max((size_t)16 * ((size_t)nkb * 16 + 16) * 2, (size_t)(4 * 16 * 17 + 16 * 17 + 16 * 40 + 16 * 96) * 4);
        if (a.KX == 1024) hipLaunchKernelGGL((k_lstm_b<false, 96, 64>), g, dim3(256), lds, s, a);
        else if (a.sh_part && a.sh_kc == 1152) hipLaunchKernelGGL((k_lstm_b<true>), g, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((k_lstm_b<false>), g, dim3(256), lds, s, a);
        return true;
    }
    if (!a.wf || (a.H & 15) || (a.KX & 15)) return false;
    if (off && !a.x_ln_g && !a.x_act) return false;  // the chunked kernel has no input transform
    const int nkb = (a.KX + a.H) / 16;
    dim3 grid(a.H / 16, (a.B + 31) / 32), block(256);
    const size_t lds = (size_t)32 * ((size_t)nkb * 16 + LSTM_PAD) * 4;
    const bool xf = a.x_ln_g || a.x_act;
    if (nkb == 96 && !xf) {  // 1024 + 512 (EfficientZero conv on 64x64 observations: 8x8 latent): 16-row tiles, 98.5 KB of LDS
        if (a.gelu && a.KX == 1024 && !getenv("LZ_LSTM_NOSPLIT"))
            hipLaunchKernelGGL((k_lstm2<96, 0, 16, 64, false, true, 36, true>), dim3(a.H / 16, (a.B + 15) / 16), block, (size_t)16 * (1024 + LSTM_PAD) * 4, s, a);
        else if (a.gelu) hipLaunchKernelGGL((k_lstm2<96, 0, 16, 0, false, true>), dim3(a.H / 16, (a.B + 15) / 16), block, (size_t)16 * ((size_t)nkb * 16 + LSTM_PAD) * 4, s, a);
        else if (a.sh_part && a.H == 512 && a.sh_kc == 2048) {   // split heads on the 8x8 latent (round 6): the head MLPs' first layers ride on this launch
            static const char *noovl = getenv("LZ_LSTM_NO_OVL");   // A/B: the h columns beside the x columns (99 KB of LDS, one workgroup per CU)
            if (noovl) hipLaunchKernelGGL((k_lstm2<96, 0, 16, 64, true, false, 64, false>), dim3(a.H / 16, (a.B + 15) / 16), block, (size_t)16 * ((size_t)nkb * 16 + LSTM_PAD) * 4, s, a);
            else hipLaunchKernelGGL((k_lstm2<96, 0, 16, 64, true, false, 64, true>), dim3(a.H / 16, (a.B + 15) / 16), block, (size_t)16 * (1024 + LSTM_PAD) * 4, s, a);
        } else if (a.KX == 1024 && !getenv("LZ_LSTM_NOSPLIT"))   // x columns first, the h columns arrive under their products and take the x columns' place in LDS
            hipLaunchKernelGGL((k_lstm2<96, 0, 16, 64, false, false, 36, true>), dim3(a.H / 16, (a.B + 15) / 16), block, (size_t)16 * (1024 + LSTM_PAD) * 4, s, a);
        else hipLaunchKernelGGL((k_lstm2<96, 0, 16>), dim3(a.H / 16, (a.B + 15) / 16), block, (size_t)16 * ((size_t)nkb * 16 + LSTM_PAD) * 4, s, a);
        return true;
    }
    if (nkb == 68 && !xf && a.gelu && a.KX == 576) {   // the same LSTM with GELU behind its BatchNorm (conv Sampled EfficientZero, 6x6 latent)
        hipLaunchKernelGGL((k_lstm2<68, 0, 16, 36, false, true>), dim3(a.H / 16, (a.B + 15) / 16), block, (size_t)16 * ((size_t)nkb * 16 + LSTM_PAD) * 4, s, a);
        return true;
    }
    // 16-row tiles (70 KB of LDS, 512 workgroups at 256 roots): two workgroups per CU, so one's staging and cell epilogue run
    // under the other's MFMAs; measured 0.6 us per launch faster than 32-row tiles (LZ_LSTM_ROWS32=1) although the gate weights
    // are streamed twice as often
    static const char *big_rows = getenv("LZ_LSTM_ROWS32");
    static const char *lstm3 = getenv("LZ_LSTM3");   // the pipelined 32-row kernel (A/B switch while it is being qualified)
    if (nkb == 68 && !xf && !big_rows && lstm3 && a.KX == 576 && a.H == 512 && a.B >= 32) {
        const dim3 g3(a.H / 16, (a.B + 31) / 32);
        const size_t l3 = (size_t)32 * ((size_t)nkb * 16 + LSTM_PAD) * 4;
        if (a.sh_part && a.sh_kc == 1152) hipLaunchKernelGGL((k_lstm3<68, true>), g3, block, l3, s, a);
        else hipLaunchKernelGGL((k_lstm3<68, false>), g3, block, l3, s, a);
        return true;
    }
    if (nkb == 68 && !xf && !big_rows) {
        static const char *nosplit = getenv("LZ_LSTM_NOSPLIT");  // the one-burst staging (A/B timing, parity: both forms are bit-identical)
        if (a.KX == 576 && !nosplit && a.sh_part && a.H == 512 && a.sh_kc == 1152)
            hipLaunchKernelGGL((k_lstm2<68, 0, 16, 36, true>), dim3(a.H / 16, (a.B + 15) / 16), block, (size_t)16 * ((size_t)nkb * 16 + LSTM_PAD) * 4, s, a);
        else if (a.KX == 576 && !nosplit) hipLaunchKernelGGL((k_lstm2<68, 0, 16, 36>), dim3(a.H / 16, (a.B + 15) / 16), block, (size_t)16 * ((size_t)nkb * 16 + LSTM_PAD) * 4, s, a);
        else hipLaunchKernelGGL((k_lstm2<68, 0, 16>), dim3(a.H / 16, (a.B + 15) / 16), block, (size_t)16 * ((size_t)nkb * 16 + LSTM_PAD) * 4, s, a);
        return true;
    }
    if (nkb == 41 && !xf) { hipLaunchKernelGGL((k_lstm2<41>), grid, block, lds, s, a); return true; }   // 16 x 9 + 512 (TicTacToe EfficientZero)
    if ((nkb == 74 || nkb == 113) && !xf) {   // 16 x 42 + 512 (6x7 boards), 16 x 81 + 512 (9x9 boards): 16-row tiles (116 KB of LDS at 9x9)
        const dim3 g16(a.H / 16, (a.B + 15) / 16);
        const size_t l16 = (size_t)16 * ((size_t)nkb * 16 + LSTM_PAD) * 4;
        if (nkb == 74) hipLaunchKernelGGL((k_lstm2<74, 0, 16>), g16, block, l16, s, a);
        else hipLaunchKernelGGL((k_lstm2<113, 0, 16>), g16, block, l16, s, a);
        return true;
    }
    if (nkb == 68 && !xf) hipLaunchKernelGGL((k_lstm2<68>), grid, block, lds, s, a);       // 576 + 512 (EfficientZero conv)
    else if (nkb == 48 && a.KX == 256) {                                                    // 256 + 512 (MLP models)
        if (xf) hipLaunchKernelGGL((k_lstm2<48, 8>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((k_lstm2<48>), grid, block, lds, s, a);
    } else if (nkb == 32 && a.KX == 256) {                                                  // 256 + 256 (the reference's LunarLander / BipedalWalker /
        if (xf) hipLaunchKernelGGL((k_lstm2<32, 8>), grid, block, lds, s, a);               //  MuJoCo / MiniGrid EfficientZero configs)
        else hipLaunchKernelGGL((k_lstm2<32>), grid, block, lds, s, a);
    } else if (nkb == 16 && a.KX == 128) {                                                  // 128 + 128
        if (xf) hipLaunchKernelGGL((k_lstm2<16, 4>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((k_lstm2<16>), grid, block, lds, s, a);
    } else return false;
    return true;
}

// One launch for (the LSTM launch `la` of the previous simulation, the tree-fused chain launch `ca` / `step` of this one): k_sim_fused.
// false: not this kernel's shape -- the caller launches the two separately.  ctl: lz_fused_ctl_bytes(...) of device memory, zeroed once per
// search; launch = 0, 1, ... within that search.
size_t lz_fused_ctl_bytes(int B, int nlaunches) { return sizeof(lz_res_ctl) + (size_t)(B / 16) * nlaunches * sizeof(unsigned); }
bool lz_launch_sim_fused(const lz_lstm_args &la, const lz_chain_args &ca, const lz_tree_step &step, void *ctl, int launch, hipStream_t s)
{
    if (launch < 0 || !ctl) return false;
    if (!(ca.gw == 6 && ca.gh == 6 && (ca.C == 0 || ca.C == 64)) || ca.gelu || ca.tstamp || ca.stamp || !ca.gather_ix || !ca.act_table || ca.nlayers <= 0) return false;
    for (int i = 0; i < ca.nlayers; ++i)
        if (!ca.layer[i].w3) return false;
    if (!(ca.B == 256 || ca.B == 128) || step.t.B != ca.B || step.t.A > 64 || step.t.variant != LZ_TREE_EFFICIENTZERO || !step.sh.on || step.ts || step.sh.dbg_logits) return false;
    if (!(la.KX == 576 && la.H == 512 && la.B == ca.B && la.sh_part && la.sh_kc == 1152 && la.wf && !la.gelu && !la.stamp && !la.x_ln_g && la.debug_hot_weights == 0)) return false;
    static int cus = -1;
    if (cus < 0) { int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0; }
    if (cus < ca.B) return false;   // one workgroup per CU, all resident at once
    const size_t lds = LZ_S3_LDS + 16;   // k_chain_s3's (>= the LSTM halves' 2 x 70 KB) + the group ids
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void *)k_sim_fused, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    lz_resident_args ra{};
    ra.ctl = reinterpret_cast<lz_res_ctl *>(ctl); ra.launch = launch; ra.ngroups = ca.B / 16; ra.B = ca.B; ra.BA = ca.B * step.t.A;
    hipLaunchKernelGGL(k_sim_fused, dim3(ca.B), dim3(512), lds, s, la, ca, step, ra);
    return true;
}

void lz_launch_lstm(const lz_lstm_args &a, hipStream_t s)
{
    if (launch_lstm2(a, s)) return;
    // 32-row tiles double the workgroup count (two resident per CU: one's chunk barrier overlaps the other's
    // MFMAs) while the batch is small; 64-row tiles halve the operand traffic once there are enough rows
#ifdef LZ_DEBUG_KNOBS
    static const char *force = getenv("LZ_DEBUG_LSTM_ROWS");
    const int rows = force ? atoi(force) : (a.B <= 512 ? 32 : 64);
#else
    const int rows = a.B <= 512 ? 32 : 64;
#endif
    if (rows == 32) launch_lstm_m<32>(a, s);
    else launch_lstm_m<64>(a, s);
}

unsigned long long *lz_debug_heads_ts = nullptr;   // timing experiments (debug build): device buffer of 8 stamps, see tools/heads_timing.py

void lz_launch_heads(const lz_head_desc *heads, int nheads, int B, int HID, hipStream_t s)
{
    head_pack hp;
    hp.ts = lz_debug_heads_ts;
    int k1max = 0;
    for (int i = 0; i < nheads && i < MAXH; ++i) {
        hp.h[i] = heads[i];
        if (heads[i].K1 > k1max) k1max = heads[i].K1;
    }
    const size_t lds = ((size_t)EPB * k1max + (size_t)EPB * HID + 8 * 8) * 4;
    static const char *narrow = getenv("LZ_HEADS_256");
    static const char *valu = getenv("LZ_HEADS_VALU");   // the VALU kernel instead of the MFMA one
    if (HID != 32) return;
    bool mm = !valu && !narrow;
    // (round 6: K1 any multiple of 16 up to 1344 -- the 9x9 / 6x7 boards' 16 x 81 / 16 x 42 head inputs ran the VALU kernel: 11.6 us per launch at 256 Go roots)
    static const char *mm64 = getenv("LZ_HEADS_MM64");   // the round-5 rule: multiples of 64 only
    for (int i = 0; i < nheads && i < MAXH; ++i) mm = mm && (heads[i].K1 & (mm64 ? 63 : 15)) == 0 && heads[i].K1 <= (mm64 ? 1024 : 1344) && heads[i].K1 >= 64 && heads[i].NOUT <= 640;
    if (mm) {
        const size_t lds2 = ((size_t)EPB * k1max + 8 * 64 + EPB * 32 + 32 + 64) * 4;
        if (k1max <= 576) hipLaunchKernelGGL(k_heads_mm<9>, dim3((B + EPB - 1) / EPB, nheads), dim3(512), lds2, s, hp, B);
        else if (k1max <= 1024) hipLaunchKernelGGL(k_heads_mm<16>, dim3((B + EPB - 1) / EPB, nheads), dim3(512), lds2, s, hp, B);
        else hipLaunchKernelGGL(k_heads_mm<21>, dim3((B + EPB - 1) / EPB, nheads), dim3(512), lds2, s, hp, B);
        return;
    }
    if (narrow) hipLaunchKernelGGL((k_heads<32, 256>), dim3((B + EPB - 1) / EPB, nheads), dim3(256), lds, s, hp, B);
    else hipLaunchKernelGGL((k_heads<32, 512>), dim3((B + EPB - 1) / EPB, nheads), dim3(512), lds, s, hp, B);
}
