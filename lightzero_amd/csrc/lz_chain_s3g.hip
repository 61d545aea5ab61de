// lz_chain_s3g.hip -- PARITY MODE (fp32 accuracy), round 6: the LDS-resident convolution chain as SPLIT-bf16 products for every latent
// grid the engine serves with 64 channels -- 8x8 (the reference's shipped Atari configuration: 64x64 observations,
// zoo/atari/config/atari_efficientzero_config.py:29,39), 9x9 (Go, BASELINE configs[3]), 6x7 (Connect4), 4x4 (2048), and the GELU networks
// of the convolutional Sampled EfficientZero on 6x6 / 8x8 (sampled_efficientzero_model.py:40).  k_chain_s3 (lz_nn.hip) stays the kernel
// of the 6x6 ReLU chain (the headline, with its split heads); this is the same arithmetic -- DESIGN.md 3.2c: every fp32 operand is
// EXACTLY the sum of three bf16 planes hi | mid | lo, six of the nine plane products per k-step on v_mfma_f32_16x16x32_bf16, each exact
// in the fp32 accumulator -- generalised in three places:
//   (1) the activations live in LDS ONLY as their three bf16 planes.  (hi + mid) + lo evaluated in fp32 IS the fp32 value, bit for bit
//       (tests/test_split_bf16_cpu.py), so the residual operand and the inputs of the 1x1 head convolutions are rebuilt from the planes and
//       the fp32 copies (4 x 17.7 KB on an 8x8 grid) are gone;
//   (2) a layer's epilogue starts behind a workgroup barrier that follows the last read of its input, so an output may overwrite its own
//       input: the launcher re-assigns the host's four logical buffers to THREE physical ones by liveness (an EfficientZero step needs two);
//   (3) two wave layouts.  Up to four 16-pixel tiles (4x4, 6x7, 8x8: no tile padding at 64 pixels) a wave owns two output-channel tiles x
//       all pixel tiles, one half of the input channels and one half of the taps (0-4 | 5-8), as in k_chain_s3: four waves hold partial sums
//       of the same tiles.  With six pixel tiles (9x9: 81 of 96 slots) 48 accumulators + 18 pixel fragments per wave do not fit 256
//       registers: the waves split the PIXELS instead (three tiles each, all nine taps), two waves share a tile's sum.
// What it replaces per grid: k_chain_w<8,8> (fp32 Winograd F(2x2,3x3): 2.25x fewer multiplications on a pipe 16x slower, plus its
// transforms), k_chain<9,9> / <7,6> / <4,4> (fp32 direct form on v_mfma_f32_16x16x4_f32).  LZ_CHAIN_NO_SPLIT=1 keeps those.
// Reference: lzero/model/efficientzero_model.py:427-569 (DynamicsNetwork), lzero/model/common.py:1081-1216 (PredictionNetwork),
// lzero/model/muzero_model.py:419-538; the residual blocks are ding's ResBlock 'basic' (conv-bn-act, conv-bn, + input, act).
#include <stdlib.h>

#include <type_traits>

#include "lz_nn_kernels.h"
#define LZ_TREE_DEV_RESTORE_FAST_CONTRACT
#include "lz_tree_dev.h"
#include "lz_nn_dev.h"

namespace {

constexpr int S3G_NBUF = 3;    // physical activation buffers (three bf16 planes each)
constexpr int S3G_PB = 80;     // bf16 per pixel of a plane: 64 + pad.  160 B = 10 bank quads: conflict-free ds_read_b128 (k_chain_s3)

// LDS bytes of an instance
constexpr int S3G_C1 = 16 * 68 + 48;   // floats of a 1x1 job in LDS: weights [16][64 + 4] (row pitch 68: conflict-free float4 reads), bias | scale | shift [16]

// LDS bytes of an instance with `nlayers` layers (the folded-BN tables stand last and are sized by the launch)
constexpr size_t s3g_lds_bytes(int hw, int nlayers)
{
    const int mt = (hw + 15) / 16;
    const bool pix = mt > 4;
    const int mtw = pix ? mt / 2 : mt, nsh = pix ? 2 : 4, ngrp = pix ? 4 : 2, nq = 2 * mtw;
    return (size_t)S3G_NBUF * 3 * (hw + 1) * S3G_PB * 2 + (size_t)ngrp * nq * (nsh - 1) * 256 * 4 + 128 * 4 + (size_t)3 * S3G_C1 * 4 + (size_t)nlayers * 128 * 4;
}

// HEADS (EfficientZero, TREE == 1): split heads -- waves 1..7 finish the head MLPs of the previous simulation's leaf from the LSTM launch's
// first-layer partials while wave 0 stages the tree (heads_in_prologue, lz_nn_dev.h): no head launch between two simulations
template <int GW, int GH, int TREE = 0, bool GELU = false, bool HEADS = false>
__global__ __launch_bounds__(512) void k_chain_s3g(lz_chain_args a, typename step_arg<TREE>::type step)
{
    static_assert(!HEADS || TREE == 1, "split heads ride on the EfficientZero tree step");
    constexpr int NW = 8, HW = GW * GH, MT = (HW + 15) / 16, NTHR = NW * 64, PB = S3G_PB, NPL = 3;
    constexpr bool PIX = MT > 4;                         // the pixel-split layout
    static_assert(!PIX || (MT % 2) == 0, "pixel halves");
    constexpr int MTW = PIX ? MT / 2 : MT;               // 16-pixel tiles per wave
    constexpr int NSH = PIX ? 2 : 4;                     // waves that hold partial sums of the same output tiles
    constexpr int NGRP = NW / NSH;                       // groups of such waves
    constexpr int NQ = 2 * MTW;                          // output tiles per wave: q = ntl MTW + mtl (channel tile 2 np + ntl, pixel tile mt0 + mtl)
    constexpr int NFIN = (NQ + NSH - 1) / NSH;           // tiles a wave finishes, at most: tile q is finished by the wave with widx == q % NSH
    constexpr int BB = (HW + 1) * PB, BB3 = NPL * BB;    // one plane, one buffer (bf16 elements); pixel HW of every plane is the all-zero halo pixel
    static_assert(MTW * 9 <= 64, "one validity bit per (tile, tap)");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16 *sB = reinterpret_cast<__bf16 *>(smem);                               // [S3G_NBUF][3 planes][HW + 1][PB]; the staged tree first
    float *sP = reinterpret_cast<float *>(sB + S3G_NBUF * BB3);                  // [NGRP][NQ][NSH - 1][64 lanes][4] partial sums
    float *sMisc = sP + NGRP * NQ * (NSH - 1) * 256;                             // 128 floats: the tree step's selection
    float *sC1 = sMisc + 128;                                                    // [3 jobs][S3G_C1]: the 1x1 head convolutions' weights and epilogue operands
    float *sSS = sC1 + 3 * S3G_C1;                                               // [nlayers][2][64] folded-BN scale | shift
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, np = wv & 1, kh = (wv >> 1) & 1, g2 = wv >> 2;
    const int widx = PIX ? kh : kh + 2 * g2;             // rank among the waves that share this wave's tiles (the fixed order of their sum)
    const int grp = PIX ? np * 2 + g2 : np;
    const int mt0 = PIX ? g2 * MTW : 0;                  // first pixel tile of this wave
    const int b = blockIdx.x;
    lz_stamp_begin(a.stamp);
#ifdef LZ_DEBUG_KNOBS   // timing experiments (tools/bench_conv_configs.py --chain-ts): cycle stamps of workgroup 0 / wave 0
    int nts = 0;
#define S3G_TS() do { if (a.tstamp && b == 0 && tid == 0) { lz_stamp_store(a.tstamp + 1 + nts, __builtin_readcyclecounter()); ++nts; } } while (0)
#define S3G_TSF(slot) do { if (a.tstamp && b == 0 && tid == 0) lz_stamp_store(a.tstamp + (slot), __builtin_readcyclecounter()); } while (0)
#else
#define S3G_TS() do { } while (0)
#define S3G_TSF(slot) do { } while (0)
#endif
    S3G_TS();
    // per-layer parameters live in lanes (lane L: layer L) and are handed out by v_readlane: a scalar load from the argument block at the top
    // of a layer is a round trip the layer waits for
    const lz_chain_layer &myl = a.layer[min(lane, LZ_CHAIN_MAX_LAYERS - 1)];
    const unsigned long long my_wb = (unsigned long long)myl.w3, my_gout = (unsigned long long)myl.gout;
    const int my_flags = myl.in | (myl.out << 2) | ((myl.res + 1) << 4) | ((myl.relu & 3) << 7) | ((myl.act != 0) << 9);
    auto lane64 = [&](unsigned long long v, int L) {
        const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, L), hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), L);
        return ((unsigned long long)hi << 32) | lo;
    };
    // the same for every pointer the prologue's staging loops and the 1x1 units follow (lane j < 3: job j): fetched through a per-thread index
    // they were a vector load from the argument segment IN FRONT of each data load -- two dependent round trips behind the latent in the
    // prologue, one per unit in the tail (stamps: 5 k cycles for two 1x1 units)
    const unsigned long long my_scale = (unsigned long long)myl.scale, my_shift = (unsigned long long)myl.shift;
    const lz_c1_job &myj = a.c1[min(lane, 2)];
    const unsigned long long my_c1w = (unsigned long long)myj.w, my_c1bias = (unsigned long long)myj.bias, my_c1scale = (unsigned long long)myj.scale,
                             my_c1shift = (unsigned long long)myj.shift, my_c1out = (unsigned long long)myj.out;
    const int my_c1stride = myj.out_stride, my_c1off = myj.out_off, my_c1misc = (myj.act & 3) | (a.c1_in[min(lane, 2)] << 2);
    // folded-BN tables -> LDS: 128 floats per layer; a wave's 64 threads stay inside one layer's scale or shift half
    // (request and store are separate steps: without a tree step the requests go out BEFORE the wait for the selection, the stores come behind
    // the latent's -- in one loop each table was a round trip of its own behind the latent, 1.1 k cycles each in the stamps)
    constexpr int NSS = (LZ_CHAIN_MAX_LAYERS * 128 + NTHR - 65) / (NTHR - 64);
    auto request_ss = [&](int first, int nthr, float (&v)[NSS]) {
#pragma unroll
        for (int k = 0; k < NSS; ++k) {
            const int i = tid - first + k * nthr;
            v[k] = 0.0f;
            if (i < a.nlayers * 128) {
                const int L = __builtin_amdgcn_readfirstlane(i >> 7), r = i & 127;
                const float *sp = reinterpret_cast<const float *>(lane64(__builtin_amdgcn_readfirstlane(r >> 6) ? my_shift : my_scale, L));
                v[k] = sp[r & 63];
            }
        }
    };
    auto store_ss = [&](int first, int nthr, const float (&v)[NSS]) {
#pragma unroll
        for (int k = 0; k < NSS; ++k) {
            const int i = tid - first + k * nthr;
            if (i < a.nlayers * 128) sSS[i] = v[k];
        }
    };
    // weights: [layer][kh][channel tile][tap][plane][64 lanes][8 bf16] (Builder::split3_chain).  A wave's fragments of a layer stream through a
    // ring of RT taps; a slot is refilled right behind the products that read it -- with this layer's tap RT further on, or with the NEXT
    // layer's first taps (so the stream runs through the partial-sum exchange and the epilogue)
    constexpr int RT = 2;
    const int t0 = PIX ? 0 : (g2 ? 5 : 0);
    const size_t wofs0 = (((size_t)kh * 4 + 2 * np) * 9) * NPL * 64 + lane, wofs1 = wofs0 + (size_t)9 * NPL * 64;
    bf16x8 wr[RT][2][NPL];
    auto load_w = [&](gbl_bf16x8 *wl, int t, int pl, bf16x8 (&dst)[2][NPL]) {
        gbl_bf16x8 *w0 = wl + (size_t)(t * NPL + pl) * 64;
        dst[0][pl] = w0[wofs0];
        dst[1][pl] = w0[wofs1];
    };
    auto load_w0 = [&]() {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) load_w(as_global_bf16x8((unsigned long long)a.layer[0].w3), t0 + i, pl, wr[i]);
    };
    // L2 keeps nothing across a kernel boundary: the workgroups of an XCD (blockIdx % 8) each touch a share of every layer's lines early,
    // results unused, so that the layers find them in L2 (k_chain_s3)
    auto prefetch_weights = [&](int L) {
        const int nr = min(max((int)gridDim.x >> 3, 1), 32), r = (b >> 3) % nr;
        constexpr int LINES = 2 * 4 * 9 * NPL * 64 * 16 / 128;
        const char *w = reinterpret_cast<const char *>(a.layer[L].w3);
        for (int ln = r + lane * nr; ln < LINES; ln += 64 * nr) (void)*reinterpret_cast<const volatile int *>(w + (size_t)ln * 128);
    };
    // the 1x1 jobs' parameters -> LDS (requested in the tail they were an exposed L2 round trip per unit: 6.9 k of 71 k cycles on the 8x8 grid)
    auto request_c1 = [&](int first, float4 (&v)[3]) {          // (threads first .. first + 267 of every job's 268 float4 pieces: 256 of the weights + 12 of bias | scale | shift)
        const int r = tid - first;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float *pw = reinterpret_cast<const float *>(lane64(my_c1w, j)), *pb = reinterpret_cast<const float *>(lane64(my_c1bias, j)),
                        *ps = reinterpret_cast<const float *>(lane64(my_c1scale, j)), *pt = reinterpret_cast<const float *>(lane64(my_c1shift, j));
            const int q = max(r - 256, 0), which = q >> 2;
            const float *src = r < 256 ? pw + max(r, 0) * 4 : (which == 0 ? pb : which == 1 ? ps : pt) + (q & 3) * 4;
            v[j] = (j < a.nc1 && r >= 0 && r < 268) ? *reinterpret_cast<const float4 *>(src) : vzero4();
        }
    };
    auto store_c1 = [&](int first, const float4 (&v)[3]) {
        const int r = tid - first;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < a.nc1 && r >= 0 && r < 268)
                *reinterpret_cast<float4 *>(sC1 + j * S3G_C1 + (r < 256 ? (r >> 4) * 68 + (r & 15) * 4 : 16 * 68 + (r - 256) * 4)) = v[j];
    };
    int g_slot = 0, g_action = 0;
    float ssv[NSS];
    float4 c1v[3];
    if constexpr (TREE == 0) {
        // requests in the order of their use: the selection (the latent gather waits for it), the first weight fragments, the tables, the L2 warm-up
        if (a.gather_ix) g_slot = a.gather_ix[b];
        if (a.act_table) g_action = a.action[b];
        load_w0();
        request_ss(0, NTHR, ssv);
        request_c1(0, c1v);
        for (int L = wv; L < a.nlayers; L += NW) prefetch_weights(L);
    }
    if constexpr (TREE != 0) {
        int32_t *s_sel = reinterpret_cast<int32_t *>(sMisc + 120);
        // split heads: the partial-sum exchange is free until the first layer -- the leaf hand-over, the counters and the head waves' scratch
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
            if constexpr (HEADS) __builtin_amdgcn_s_setprio(3);   // one wave of strictly dependent instructions: first wherever it competes with the head waves
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
            load_w0();
            for (int L = wv - 1; L < a.nlayers; L += NW - 1) prefetch_weights(L);
            request_ss(64, NTHR - 64, ssv);
            request_c1(64, c1v);
            store_ss(64, NTHR - 64, ssv);
            store_c1(64, c1v);
        }
        __syncthreads();
        g_slot = s_sel[0];
        g_action = s_sel[1];
        if (wv == 0) load_w0();
    }
    // geometry of the tiles this wave FINISHES (f-th: q = widx + NSH f), the same for every layer
    int mpix[NFIN], c4v[NFIN];
#pragma unroll
    for (int f = 0; f < NFIN; ++f) {
        const int q = min(widx + NSH * f, NQ - 1), ntl = q / MTW, mtl = q - MTW * ntl;
        mpix[f] = (mt0 + mtl) * 16 + (lane & 15);
        c4v[f] = (2 * np + ntl) * 16 + 4 * (lane >> 4);
    }
    // the action table's rows of those tiles (the dynamics convolution adds them): requested with the latent, so that the layers' only
    // vector-memory traffic is the weight stream and no wait of theirs has to drain it
    f32x4 tvv[NFIN];
#pragma unroll
    for (int f = 0; f < NFIN; ++f) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        tvv[f] = a.act_table ? *reinterpret_cast<const f32x4 *>(a.act_table + (size_t)g_action * HW * 64 + min(mpix[f], HW - 1) * 64 + c4v[f]) : z;
    }
    {
        // the latent: fp32 NHWC rows from the pool -> the three planes of the first layer's input buffer
        const int in0 = __builtin_amdgcn_readlane(my_flags, 0) & 3;
        __bf16 *dst = sB + in0 * BB3;
        const float *src = a.in + (size_t)b * HW * 64 + (size_t)g_slot * a.slot_stride;
        constexpr int NU = (HW * 16 + NTHR - 1) / NTHR;
        float4 v[NU];
        S3G_TSF(50);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int idx = min(u * NTHR + tid, HW * 16 - 1);
            v[u] = *reinterpret_cast<const float4 *>(src + (size_t)idx * 4);
        }
        S3G_TSF(51);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int idx = u * NTHR + tid;
            if (idx < HW * 16) {
                bf16x4 h, m3, l3;
                split3_bf16((f32x4){v[u].x, v[u].y, v[u].z, v[u].w}, h, m3, l3);
                *reinterpret_cast<bf16x4 *>(dst + (idx >> 4) * PB + (idx & 15) * 4) = h;
                *reinterpret_cast<bf16x4 *>(dst + BB + (idx >> 4) * PB + (idx & 15) * 4) = m3;
                *reinterpret_cast<bf16x4 *>(dst + 2 * BB + (idx >> 4) * PB + (idx & 15) * 4) = l3;
            }
        }
        S3G_TSF(52);
        // the all-zero halo pixel of every plane of every buffer
        if (tid < S3G_NBUF * NPL * (PB / 8)) *reinterpret_cast<float4 *>(sB + (tid / (PB / 8)) * BB + HW * PB + (tid % (PB / 8)) * 8) = vzero4();
        if (TREE == 0) {   // behind the latent: nothing in front of the first layer waits for these
            S3G_TSF(53);
            store_ss(0, NTHR, ssv);
            S3G_TSF(54);
            store_c1(0, c1v);
            S3G_TSF(55);
        }
    }
    // ---- per-lane geometry of the products: B columns of this lane = pixels 16 (mt0 + mtl) + (lane & 15); tap (dy, dx) reads pixel
    // m + dy GW + dx when it is inside the image, the zero pixel otherwise (one validity bit per (tile, tap))
    unsigned long long valid = 0;
    int abase[MTW];
#pragma unroll
    for (int mtl = 0; mtl < MTW; ++mtl) {
        const int m = (mt0 + mtl) * 16 + (lane & 15), y = m / GW, x = m - y * GW;
        abase[mtl] = (min(m, HW) * PB + kh * 32 + (lane >> 4) * 8) * 2;   // bytes
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            if (m < HW && yy >= 0 && yy < GH && xx >= 0 && xx < GW) valid |= 1ull << (mtl * 9 + t);
        }
    }
    const int azero = (HW * PB + kh * 32 + (lane >> 4) * 8) * 2;
    S3G_TS();
    __syncthreads();
    S3G_TS();

    // The MFMAs run TRANSPOSED (weights as the A operand): D[channel][pixel] -- a lane ends up with four consecutive channels of ONE pixel,
    // 8 contiguous bytes in every plane
    const int nlayers = a.nlayers;
    for (int L = 0; L < nlayers; ++L) {
        const int flags = __builtin_amdgcn_readlane(my_flags, L);
        const int Ln = L + 1 < nlayers ? L + 1 : L;
        const char *sBin = reinterpret_cast<const char *>(sB + (flags & 3) * BB3);
        __bf16 *sBout = sB + ((flags >> 2) & 3) * BB3;
        gbl_bf16x8 *wl_cur = as_global_bf16x8(lane64(my_wb, L)), *wl_nxt = as_global_bf16x8(lane64(my_wb, Ln));
        const bool tab = (flags >> 9) & 1;
        const int actc = (flags >> 7) & 3;
        const int res = ((flags >> 4) & 7) - 1;
        // operands of the epilogue that do not depend on the products -- folded-BN scale / shift and the residual of the tiles this wave
        // finishes, rebuilt from its planes -- are read here: the LDS pipe has room under the products
        const __bf16 *sRes = sB + max(res, 0) * BB3;
        f32x4 rvv[NFIN], scv[NFIN], shv[NFIN];
#pragma unroll
        for (int f = 0; f < NFIN; ++f) {
            const int o = min(mpix[f], HW) * PB + c4v[f];
            const bf16x4 rh = *reinterpret_cast<const bf16x4 *>(sRes + o), rm = *reinterpret_cast<const bf16x4 *>(sRes + BB + o),
                         rl = *reinterpret_cast<const bf16x4 *>(sRes + 2 * BB + o);
#pragma unroll
            for (int c = 0; c < 4; ++c) rvv[f][c] = res >= 0 ? ((float)rh[c] + (float)rm[c]) + (float)rl[c] : 0.0f;
            scv[f] = *reinterpret_cast<const f32x4 *>(sSS + L * 128 + c4v[f]);
            shv[f] = *reinterpret_cast<const f32x4 *>(sSS + L * 128 + 64 + c4v[f]);
        }
        auto read_x = [&](int t, int pl, bf16x8 (&af)[NPL][MTW]) {
            const int toff = ((t / 3 - 1) * GW + (t % 3 - 1)) * PB * 2;
#pragma unroll
            for (int mtl = 0; mtl < MTW; ++mtl) {
                const int off = (valid >> (mtl * 9 + t)) & 1 ? abase[mtl] + toff : azero;
                af[pl][mtl] = *reinterpret_cast<const bf16x8 *>(sBin + off + pl * BB * 2);
            }
        };
        f32x4 acc[2][MTW];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int mtl = 0; mtl < MTW; ++mtl) acc[n][mtl] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // One tap = six of the nine plane products (hi mid lo = planes 0 1 2), 2 MTW MFMAs each, consecutive MFMAs writing different
        // accumulators.  The order of the products frees x[lo] after the first, w[hi] after the third, x[mid] after the fourth, w[mid] after
        // the fifth; each is re-requested right there (k_chain_s3's schedule)
        auto prod = [&](const bf16x8 (&w)[2][NPL], int wp, const bf16x8 (&x)[NPL][MTW], int xp) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int mtl = 0; mtl < MTW; ++mtl) acc[n][mtl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[n][wp], x[xp][mtl], acc[n][mtl], 0, 0, 0);
        };
        auto taps = [&](auto ntaps_c, auto tb_c) {
            constexpr int NTAP = decltype(ntaps_c)::value, tb = decltype(tb_c)::value;
            bf16x8 x[NPL][MTW];
#pragma unroll
            for (int pl = NPL - 1; pl >= 0; --pl) read_x(tb, pl, x);
#pragma unroll
            for (int i = 0; i < NTAP; ++i) {
                bf16x8 (&w)[2][NPL] = wr[i % RT];
                const bool more = i + 1 < NTAP;
                gbl_bf16x8 *wl = (i + RT < NTAP) ? wl_cur : wl_nxt;
                const int wt = (i + RT < NTAP) ? tb + i + RT : tb + i % RT;
                prod(w, 0, x, 2);                           // hi  x lo
                __builtin_amdgcn_sched_barrier(0);
                if (more) read_x(tb + i + 1, 2, x);
                prod(w, 0, x, 1);                           // hi  x mid
                prod(w, 0, x, 0);                           // hi  x hi
                __builtin_amdgcn_sched_barrier(0);
                load_w(wl, wt, 0, w);
                prod(w, 1, x, 1);                           // mid x mid
                __builtin_amdgcn_sched_barrier(0);
                if (more) read_x(tb + i + 1, 1, x);
                prod(w, 1, x, 0);                           // mid x hi
                __builtin_amdgcn_sched_barrier(0);
                load_w(wl, wt, 1, w);
                prod(w, 2, x, 0);                           // lo  x hi
                __builtin_amdgcn_sched_barrier(0);
                load_w(wl, wt, 2, w);
                if (more) read_x(tb + i + 1, 0, x);
            }
        };
#ifdef LZ_DEBUG_KNOBS   // per-wave start / end of the products of layer 2 (slots 32.., 40..)
        if (a.tstamp && b == 0 && lane == 0 && L == 2) lz_stamp_store(a.tstamp + 32 + wv, __builtin_readcyclecounter());
#endif
        if constexpr (PIX) {
            taps(std::integral_constant<int, 9>{}, std::integral_constant<int, 0>{});
        } else {
            if (g2 == 0) taps(std::integral_constant<int, 5>{}, std::integral_constant<int, 0>{});
            else taps(std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{});
        }
#ifdef LZ_DEBUG_KNOBS
        if (a.tstamp && b == 0 && lane == 0 && L == 2) lz_stamp_store(a.tstamp + 40 + wv, __builtin_readcyclecounter());
#endif
        S3G_TS();
        // ---- the partial sums of an output tile meet in LDS: every wave leaves the tiles it does not finish
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int fin = q % NSH;
            if (fin != widx) {
                const int rank = widx < fin ? widx : widx - 1;
                *reinterpret_cast<f32x4 *>(sP + (((grp * NQ + q) * (NSH - 1) + rank) * 64 + lane) * 4) = acc[q / MTW][q % MTW];
            }
        }
        __syncthreads();
        S3G_TS();
        {
            float *go = reinterpret_cast<float *>(lane64(my_gout, L));
            if (go) go += (size_t)b * HW * 64;
            // the wave's role as a compile-time constant behind a scalar branch (run-time accumulator picks are v_cndmask chains); the
            // action-table rows enter by an FMA with 0 | 1, ReLU is a max with 0 | -inf
            const float tabf = tab ? 1.0f : 0.0f, floor_ = actc == 1 ? 0.0f : -__builtin_inff();
            auto finish = [&](auto widx_c) {
                constexpr int W = decltype(widx_c)::value;
                constexpr int NF = (NQ - W + NSH - 1) / NSH;         // tiles W, W + NSH, ... < NQ
                if constexpr (NF > 0) {
                    // every LDS read of the epilogue before its first write (the compiler must assume they alias)
                    f32x4 oth[NF][NSH - 1];
#pragma unroll
                    for (int f = 0; f < NF; ++f)
#pragma unroll
                        for (int r = 0; r < NSH - 1; ++r) oth[f][r] = *reinterpret_cast<const f32x4 *>(sP + (((grp * NQ + W + NSH * f) * (NSH - 1) + r) * 64 + lane) * 4);
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        const int q = W + NSH * f;
                        // the partial sums in the fixed order of the waves' ranks: the finisher's own stands at position W
                        f32x4 p[NSH];
#pragma unroll
                        for (int w4 = 0; w4 < NSH; ++w4) p[w4] = (w4 == W) ? acc[q / MTW][q % MTW] : oth[f][w4 < W ? w4 : w4 - 1];
                        f32x4 o;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            float v = p[0][c] + p[1][c];
                            if constexpr (NSH == 4) v = (v + p[2][c]) + p[3][c];
                            v = __builtin_fmaf(tabf, tvv[f][c], v);
                            v = v * scv[f][c] + shv[f][c];
                            v += rvv[f][c];
                            if constexpr (GELU) o[c] = actc == 2 ? gelu_tanh_(v) : fmaxf(v, floor_);
                            else o[c] = fmaxf(v, floor_);
                        }
                        bf16x4 oh, om, ol;
                        split3_bf16(o, oh, om, ol);
                        if (mpix[f] < HW) {
                            *reinterpret_cast<bf16x4 *>(sBout + mpix[f] * PB + c4v[f]) = oh;
                            *reinterpret_cast<bf16x4 *>(sBout + BB + mpix[f] * PB + c4v[f]) = om;
                            *reinterpret_cast<bf16x4 *>(sBout + 2 * BB + mpix[f] * PB + c4v[f]) = ol;
                            if (go) store_wt(go + mpix[f] * 64 + c4v[f], o);
                        }
                    }
                }
            };
            if constexpr (NSH == 4) {
                switch (__builtin_amdgcn_readfirstlane(widx)) {
                case 0: finish(std::integral_constant<int, 0>{}); break;
                case 1: finish(std::integral_constant<int, 1>{}); break;
                case 2: finish(std::integral_constant<int, 2>{}); break;
                default: finish(std::integral_constant<int, 3>{}); break;
                }
            } else {
                if (__builtin_amdgcn_readfirstlane(widx) == 0) finish(std::integral_constant<int, 0>{});
                else finish(std::integral_constant<int, 1>{});
            }
        }
        S3G_TS();
        __syncthreads();
        S3G_TS();
    }
    // ---- 1x1 head convolutions (64 -> 16) + bias + BN + activation in fp32 on v_mfma_f32_16x16x4_f32; unit = (job, 16-pixel tile), wave wv runs
    // units wv, wv + 8, ...; the input rows are rebuilt from their planes, the parameters come from LDS
    const int kq4 = (lane >> 4) * 4;
    // (the unit index as a SCALAR: the job descriptors then come by s_load instead of a vector load from the argument segment per field)
    for (int u = __builtin_amdgcn_readfirstlane(wv); u < a.nc1 * MT; u += NW) {
        const int job = u / MT, i = u - job * MT;
        const int misc = __builtin_amdgcn_readlane(my_c1misc, job);
        const float *pc = sC1 + job * S3G_C1;
        float4 c1w[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) c1w[g] = *reinterpret_cast<const float4 *>(pc + (lane & 15) * 68 + g * 16 + kq4);
        const float4 c1b = *reinterpret_cast<const float4 *>(pc + 16 * 68 + kq4), c1s = *reinterpret_cast<const float4 *>(pc + 16 * 68 + 16 + kq4),
                     c1t = *reinterpret_cast<const float4 *>(pc + 16 * 68 + 32 + kq4);
        const __bf16 *sIn = sB + (misc >> 2) * BB3;
        const int row = i * 16 + (lane & 15);
        const int off = min(row, HW) * PB + kq4;
        bf16x4 xh[4], xm[4], xl[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            xh[g] = *reinterpret_cast<const bf16x4 *>(sIn + off + g * 16);
            xm[g] = *reinterpret_cast<const bf16x4 *>(sIn + BB + off + g * 16);
            xl[g] = *reinterpret_cast<const bf16x4 *>(sIn + 2 * BB + off + g * 16);
        }
        // two accumulators (k = 0..31 | 32..63, added at the end): sixteen dependent MFMAs were the longest chain of this phase
        if (u < NW) S3G_TSF(56);
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, acc2 = acc;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xf = ((float)xh[g][j] + (float)xm[g][j]) + (float)xl[g][j];
                const float xf2 = ((float)xh[g + 2][j] + (float)xm[g + 2][j]) + (float)xl[g + 2][j];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vget(c1w[g], j), xf, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(vget(c1w[g + 2], j), xf2, acc2, 0, 0, 0);
            }
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] += acc2[c];
        if (u < NW) S3G_TSF(57);
        if (row < HW) {
            f32x4 v;
            v[0] = (acc[0] + c1b.x) * c1s.x + c1t.x;
            v[1] = (acc[1] + c1b.y) * c1s.y + c1t.y;
            v[2] = (acc[2] + c1b.z) * c1s.z + c1t.z;
            v[3] = (acc[3] + c1b.w) * c1s.w + c1t.w;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if constexpr (GELU) v[c] = (misc & 3) == 2 ? gelu_tanh_(v[c]) : fmaxf(v[c], 0.0f);
                else v[c] = fmaxf(v[c], 0.0f);
            }
            store_wt(reinterpret_cast<float *>(lane64(my_c1out, job)) + ((size_t)b * HW + row) * __builtin_amdgcn_readlane(my_c1stride, job) +
                         __builtin_amdgcn_readlane(my_c1off, job) + kq4, v);
        }
        if (u < NW) S3G_TSF(58);
    }
    S3G_TS();
#ifdef LZ_DEBUG_KNOBS
    if (a.tstamp && b == 0 && tid == 0) lz_stamp_store(a.tstamp, (unsigned long long)nts);
#endif
#undef S3G_TS
#undef S3G_TSF
    lz_stamp_end(a.stamp);
}

// The host's chain uses up to four logical LDS buffers (lz_chain_layer::in / out / res, lz_chain_args::c1_in).  Values are re-assigned to
// S3G_NBUF physical buffers by liveness; a layer's output may take the buffer of a value whose last read is that very layer (its input
// or its residual): the epilogue writes behind the barrier that follows the layer's last read.  false: does not fit.
bool s3g_assign_buffers(lz_chain_args &a)
{
    const int n = a.nlayers;
    int val_of[4] = {-1, -1, -1, -1};            // logical buffer -> value id held (0: the latent; L + 1: output of layer L)
    int last_read[LZ_CHAIN_MAX_LAYERS + 1];
    for (int i = 0; i <= n; ++i) last_read[i] = -1;
    val_of[a.layer[0].in & 3] = 0;
    int vin[LZ_CHAIN_MAX_LAYERS], vres[LZ_CHAIN_MAX_LAYERS], vc1[3] = {-1, -1, -1};
    for (int L = 0; L < n; ++L) {
        const lz_chain_layer &l = a.layer[L];
        if (l.in < 0 || l.in > 3 || l.out < 0 || l.out > 3 || l.res > 3) return false;
        vin[L] = val_of[l.in];
        vres[L] = l.res >= 0 ? val_of[l.res] : -1;
        if (vin[L] < 0 || (l.res >= 0 && vres[L] < 0)) return false;
        last_read[vin[L]] = L;
        if (vres[L] >= 0) last_read[vres[L]] = L;
        val_of[l.out] = L + 1;
    }
    for (int j = 0; j < a.nc1; ++j) {
        if (a.c1_in[j] < 0 || a.c1_in[j] > 3 || val_of[a.c1_in[j]] < 0) return false;
        vc1[j] = val_of[a.c1_in[j]];
        last_read[vc1[j]] = n;
    }
    int phys[LZ_CHAIN_MAX_LAYERS + 1], holder[S3G_NBUF];
    for (int p = 0; p < S3G_NBUF; ++p) holder[p] = -1;
    phys[0] = 0; holder[0] = 0;
    for (int L = 0; L < n; ++L) {
        int pick = -1;
        for (int p = 0; p < S3G_NBUF && pick < 0; ++p)
            if (holder[p] < 0 || last_read[holder[p]] <= L) pick = p;
        if (pick < 0) return false;
        phys[L + 1] = pick; holder[pick] = L + 1;
    }
    for (int L = 0; L < n; ++L) {
        a.layer[L].in = phys[vin[L]];
        a.layer[L].res = vres[L] >= 0 ? phys[vres[L]] : -1;
        a.layer[L].out = phys[L + 1];
    }
    for (int j = 0; j < a.nc1; ++j) a.c1_in[j] = phys[vc1[j]];
    return true;
}

template <int GW, int GH, bool GELU>
void s3g_launch(const lz_chain_args &a, hipStream_t s, const lz_tree_step *step)
{
    constexpr size_t lds_max = s3g_lds_bytes(GW * GH, LZ_CHAIN_MAX_LAYERS - 1);   // 13 layers: three residual blocks per network
    static_assert(lds_max <= 160 * 1024, "LDS budget of a CU");
    const size_t lds = s3g_lds_bytes(GW * GH, a.nlayers);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void *)k_chain_s3g<GW, GH, 0, GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        if constexpr (!GELU && GW == 8) {
            (void)hipFuncSetAttribute((const void *)k_chain_s3g<GW, GH, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
            (void)hipFuncSetAttribute((const void *)k_chain_s3g<GW, GH, 1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
            (void)hipFuncSetAttribute((const void *)k_chain_s3g<GW, GH, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        }
        attr = true;
    }
    const dim3 g(a.B), blk(512);
    if constexpr (!GELU && GW == 8) {
        if (step) {
            if (step->t.variant == LZ_TREE_EFFICIENTZERO && step->sh.on) hipLaunchKernelGGL((k_chain_s3g<GW, GH, 1, false, true>), g, blk, lds, s, a, *step);
            else if (step->t.variant == LZ_TREE_EFFICIENTZERO) hipLaunchKernelGGL((k_chain_s3g<GW, GH, 1, false>), g, blk, lds, s, a, *step);
            else hipLaunchKernelGGL((k_chain_s3g<GW, GH, 2, false>), g, blk, lds, s, a, *step);
            return;
        }
    }
    hipLaunchKernelGGL((k_chain_s3g<GW, GH, 0, GELU>), g, blk, lds, s, a, no_step{});
}

}  // namespace

// grids with a k_chain_s3g instance (64 channels).  fused: the instance that runs the tree step in its prologue
bool lz_chain_s3g_supported(int gw, int gh, bool gelu, bool fused)
{
    if (fused) return !gelu && gw == 8 && gh == 8;
    if (gelu) return (gw == 6 && gh == 6) || (gw == 8 && gh == 8);
    return (gw == 8 && gh == 8) || (gw == 9 && gh == 9) || (gw == 7 && gh == 6) || (gw == 4 && gh == 4);
}

// false: not this kernel's launch (no instance, a layer without split planes, buffers that do not fit) -- the caller keeps its fp32 chains
bool lz_launch_chain_s3g(const lz_chain_args &a0, hipStream_t s, const lz_tree_step *step)
{
#ifndef LZ_DEBUG_KNOBS
    if (a0.tstamp) return false;
#endif
    if (!(a0.C == 0 || a0.C == 64) || a0.nlayers <= 0 || a0.nlayers >= LZ_CHAIN_MAX_LAYERS || a0.nc1 > 3) return false;
    if (!lz_chain_s3g_supported(a0.gw, a0.gh, a0.gelu != 0, step != nullptr)) return false;
    for (int i = 0; i < a0.nlayers; ++i)
        if (!a0.layer[i].w3) return false;
    lz_chain_args a = a0;
    if (!s3g_assign_buffers(a)) return false;
    if (a.gelu) {
        if (a.gw == 6) s3g_launch<6, 6, true>(a, s, nullptr);
        else s3g_launch<8, 8, true>(a, s, nullptr);
        return true;
    }
    if (a.gw == 8) s3g_launch<8, 8, false>(a, s, step);
    else if (a.gw == 9) s3g_launch<9, 9, false>(a, s, nullptr);
    else if (a.gw == 7) s3g_launch<7, 6, false>(a, s, nullptr);
    else s3g_launch<4, 4, false>(a, s, nullptr);
    return true;
}
