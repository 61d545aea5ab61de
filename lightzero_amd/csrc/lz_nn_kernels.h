// lz_nn_kernels.h -- fp32 network kernels for gfx950 (declarations; definitions in lz_nn.hip).
//
// All activations are NHWC fp32 in HBM.  The GEMM-shaped work (3x3 convolutions, LSTM gates) runs on
// the exact-fp32 matrix core instruction v_mfma_f32_16x16x4_f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// 3x3 convolution, pad 1, stride 1|2, as an implicit GEMM:
//   out[m][co] = epilogue( sum_{tap,ci} in[pix(m)+tap][ci] * w[co][tap][ci] )
struct lz_conv_args {
    const float *in;        // NHWC [B][Hin][Win][CIN]   (or a pool base when gather_ix != null)
    const int32_t *gather_ix;  // optional [B]: image b is read from slot gather_ix[b] of a pool
    int64_t slot_stride;    // floats between pool slots (B*Hin*Win*CIN); used only with gather_ix
    const float *w;         // packed [Cout/16][9][16][CIN]
    const float *wf;        // optional MFMA-fragment order [Cout/16][9][CIN/16][64 lanes][4] (big-grid kernel)
    const float *uf;        // optional Winograd F(2x2,3x3) weights U = G g G^T in fragment order [Cout/16][16 points][CIN/16][64 lanes][4]
    const void *wb;         // optional (fast mode): bf16 MFMA fragments [Cout/16][9 taps x CIN/32][64 lanes][8] (k_conv_bf)
    int act_bf16;           // with wb: in / residual / out are bf16 NHWC tensors (the fast tower keeps its activations in bf16), not fp32
    const void *w3;         // optional (parity mode): the weights split EXACTLY into three bf16 planes, [Cout/16][9 taps x CIN/32][3 planes][64 lanes][8]
                            // (k_conv_s3: six split-bf16 products per k-step = fp32 accuracy on the bf16 matrix pipe; activations stay fp32 tensors)
    const float *scale;     // [Cout] folded eval-mode BN scale (1 when no norm)
    const float *shift;     // [Cout]
    const float *act_table; // optional [A][Hout*Wout][Cout]: contribution of the one-hot action planes
    const int32_t *action;  // [B] (with act_table)
    const float *residual;  // optional NHWC [B][Hout][Wout][Cout]; residual_gather => read like `in`
    int residual_gather;
    float *out;             // NHWC [B][Hout][Wout][Cout]
    int B, Hin, Win, Hout, Wout, Cout;
    int relu;
};

void lz_launch_conv3x3(const lz_conv_args &a, int cin, int stride, hipStream_t s);
// two convolutions of the SAME input in one launch (the downsample block's conv1 and shortcut conv3: parity mode, 32 -> 64, stride 2);
// false = no such instance for these arguments, nothing was launched
bool lz_launch_conv3x3_pair(const lz_conv_args &a, const lz_conv_args &b, int cin, int stride, hipStream_t s);

// first layer of DownSample: conv3x3 stride 2 from NCHW obs [B][C][H][W] (C <= 4... any small C) to NHWC
void lz_launch_conv_first(const float *obs_nchw, const float *w /*[9][C][Cout]*/, const float *scale,
                          const float *shift, float *out, int B, int C, int H, int W, int Cout, hipStream_t s, int out_bf16 = 0);

// first layer of a no-downsample representation network: conv3x3 stride 1 from NCHW obs to NHWC [B][H*W][64] + BN + ReLU
void lz_launch_conv_in(const float *obs_nchw, const float *w /*[9][C][Cout]*/, const float *scale, const float *shift,
                       float *out, int B, int C, int H, int W, int Cout, hipStream_t s);

// AvgPool2d(kernel 3, stride 2, pad 1, count_include_pad) on NHWC
void lz_launch_avgpool(const float *in, float *out, int B, int Hin, int Win, int C, hipStream_t s, int in_bf16 = 0, int out_bf16 = 0);   // (bf16 tensors: fast mode)

// fast mode: n bf16 values (n a multiple of 8) -> fp32
void lz_launch_bf16_to_f32(const void *in, float *out, size_t n, hipStream_t s);

// conv1x1 (64 -> 16 channels per job) + bias + BN + ReLU on NHWC [npix][64] as a small MFMA GEMM; up to 4
// independent jobs (e.g. reward / value / policy head convolutions) share one launch (blockIdx.y = job).
struct lz_c1_job {
    const float *in;      // [npix][64]
    const float *w;       // [16][64]
    const float *bias, *scale, *shift;  // [16]
    float *out;           // out[pix * out_stride + out_off + c]
    int out_stride, out_off;
    int act;              // k_chain_w's GELU instance (lz_chain_args::gelu): 2 = GELU(tanh) instead of ReLU for this job
};
struct lz_c1_args {
    lz_c1_job job[4];
    int njobs, npix;
};
void lz_launch_conv1x1(const lz_c1_args &a, hipStream_t s);

// Chain of 3x3 convolutions on a 6x6x64 latent, one workgroup per root, activations resident in LDS across
// layers (no HBM round trip, no kernel boundary between layers), weights streamed from L2 in MFMA-fragment order.
// Used for the recurrent step (dynamics conv + residual block + prediction residual block + the 1x1 head convs)
// and for the tail of the representation network.
struct lz_chain_layer {
    const float *wf;       // fragment-packed weights [4][9][4][64][4]  (N-tile, tap, 16-channel group, lane, 4 floats)
    const float *uc;       // optional: Winograd F(2x2,3x3) weights [16 points][16 channel quads][64 = cout][4] (k_chain_w, 6x6 grids)
    const void *wb;        // optional (fast mode, lz_model_cfg::precision = 1): bf16 MFMA fragments [2 k halves][4 N-tiles][9 taps][64 lanes][8] (k_chain_b, 6x6 grids)
    const void *w3;        // optional (parity mode): the same fragments split exactly into three bf16 planes, [2][4][9 taps][3 planes][64 lanes][8] (k_chain_s3, k_chain_s3g)
    const float *scale, *shift;  // [64] folded BatchNorm
    int in, out, res;      // LDS buffer indices (0..3); res < 0: no residual
    int relu, act;         // relu: 0 none, 1 ReLU, 2 GELU(tanh) (2 only with lz_chain_args::gelu); act: add the one-hot-action table before BN (dynamics conv)
    float *gout;           // optional global NHWC [B][36][64] copy of the output (latent pool slot)
};
#define LZ_CHAIN_MAX_LAYERS 14   // dynamics conv + 2 k convs of the dynamics blocks + 2 k of the prediction blocks, num_res_blocks k <= 3
struct lz_chain_args {
    const float *in;             // NHWC [B][36][64], or a pool base when gather_ix != null
    const int32_t *gather_ix;    // optional [B] pool slot per root
    int64_t slot_stride;
    const float *act_table;      // [A][36][64]
    const int32_t *action;       // [B]
    lz_chain_layer layer[LZ_CHAIN_MAX_LAYERS];
    int nlayers;
    int C;                       // channels of every layer: 64 (k_chain, the tuned kernel), 32 or 16 (k_chain_small: board-game configs)
    lz_c1_job c1[3];             // 1x1 head convolutions on LDS buffers c1_in[j]; .in is ignored
    int c1_in[3];
    int nc1;
    int B;
    int gw, gh;                  // latent grid (6x6 Atari with downsample, 9x9 Go); compiled instances: 6x6, 9x9
    int gelu;                    // 1: some layer / 1x1 job of this launch uses GELU(tanh) (relu = 2 / act = 2): the kernel instance that reads those codes
    unsigned long long *tstamp;  // debugging: s_memtime stamps of workgroup 0 / wave 0 (null in production)
    int debug_flags;             // timing experiments of the debug build (results are then wrong): 1 = no latent gather, 2 = no action-table slice
    unsigned long long *stamp;   // optional [2]: {start of the first workgroup, end of the last workgroup (by block id)} in s_memrealtime ticks (100 MHz,
                                 // constant rate) -- bench.py's in-graph timing of the roofline kernel (k_chain_w only); null in production
};
// step != null: the tree step of every root (expand + backup of the previous simulation, selection of this one) runs as the
// prologue of the root's workgroup, and the chain reads the selected (slot, action) from LDS instead of gather_ix / action
// (which the step still writes for the kernels that follow).  Requires lz_chain_fusable(step).
struct lz_tree_step;
// The heads of the PREVIOUS simulation, finished inside the chain launch whose prologue runs that simulation's tree step: waves 1..7
// of a root's workgroup sum the first-layer partials the LSTM launch left (fixed order), apply bias / BatchNorm / ReLU, run the second
// layers (value and value prefix: softmax . support -> inverse scalar transform; policy: logits), write the leaf's network outputs
// to the pool slot AND hand them to wave 0 through LDS, which has been staging the root's tree meanwhile.
struct lz_split_heads {
    int on;
    const float *part;                    // [B][3 heads][H / 16 unit tiles][32 hidden]: first-layer partial sums left by the LSTM launch
    const float *b1[3], *s1[3], *t1[3];   // 0 value, 1 policy, 2 value prefix
    const float *w2t[3], *b2[3];          // [32 / 4][NOUT][4], [NOUT]
    int nout;                             // support size of the value / value-prefix heads (<= 768)
    int n_unit_tiles;                     // H / 16
    float support_min;
    float *out_value, *out_vp, *out_logits;   // pool slot of the leaf: [B], [B], [B][A]
    // observability for the parity tests (null unless the roots are tracing): the support-wide logits of the value (0) / value-prefix (1)
    // head [2][dbg_B][nout] and the pre-transform expectation softmax . support [2][dbg_B] of THIS simulation's leaf
    float *dbg_logits, *dbg_expect;
    int dbg_B;
};
bool lz_chain_fusable(const lz_chain_args &a, const lz_tree_step &step);
bool lz_chain_small_supported(int gw, int gh, int C);
void lz_launch_chain(const lz_chain_args &a, hipStream_t s, const lz_tree_step *step = nullptr);
// lz_chain_s3g.hip: the split-bf16 chain for the grids other than the 6x6 ReLU chain (8x8, 9x9, 6x7, 4x4; GELU networks on 6x6 / 8x8).
// lz_launch_chain_s3g returns false when the launch is not this kernel's (lz_launch_chain then keeps its fp32 chains).
bool lz_chain_s3g_supported(int gw, int gh, bool gelu, bool fused);
bool lz_launch_chain_s3g(const lz_chain_args &a, hipStream_t s, const lz_tree_step *step);

// one LSTM step (nn.LSTM, 1 layer) fused with BatchNorm1d + ReLU of the output:
//   gates = [x | h] . Wcat^T + bias ; c' = sig(f) c + sig(i) tanh(g) ; h' = sig(o) tanh(c')
struct lz_lstm_args {
    const float *x;          // [B][KX]
    const float *x_ln_g, *x_ln_b;  // optional deferred LayerNorm + activation of x's producer, applied while staging (k_lstm2 only)
    float x_ln_eps;
    int x_act;               // 0 none, 1 ReLU, 2 GELU(tanh)
    const float *h_pool, *c_pool;  // pools [NN][B][H]; row b read from slot gather_ix[b]
    const int32_t *gather_ix;      // [B]
    const float *wcat;       // [4H][KX+H], row n = 4*unit + gate (gate order i,f,g,o)
    const float *wf;         // the same matrix in MFMA-fragment order [H/16][4 gates][(KX+H)/16][64 lanes][4] (lz_lstm_pack_fragments);
                             // null => only the chunked kernel can run
    const void *wb;          // fast mode (lz_model_cfg::precision = 1; KX = 576, H = 512): bf16 fragments [H/16][4 gates][(KX+H)/32][64 lanes][8] -> k_lstm_b
    const float *bias;       // [4H] same order (b_ih + b_hh)
    const float *bn_scale, *bn_shift;  // [H]; null => hbn_out = h' (no norm / activation)
    const int32_t *search_len;  // [B] (reset when search_len % horizon == 0); may be null => no reset
    int horizon;
    float *h_out, *c_out;    // [B][H] destination slot of the pools
    float *hbn_out;          // [B][H] relu(bn(h'))
    int B, KX, H;
    int gelu;                // 1: hbn_out = gelu(bn(h')) instead of relu(bn(h')) (k_lstm2 instances of the conv Sampled EfficientZero)
    int debug_hot_weights;   // timing experiment of the debug build only (see k_lstm2); 0 in production
    // split heads (optional, sh_part != null; EfficientZero 6x6 instance only): the FIRST layers of the three head MLPs are
    // computed here as partial sums, one block per (16-row tile, unit tile u) -- the value-prefix head's over this workgroup's own 16
    // hidden units, the value / policy heads' over columns 36 u .. 36 u + 35 of the combined 1x1-conv output rows (sh_pv) -- and the
    // next chain launch finishes the heads for its root while its tree step stages (lz_split_heads)
    const float *sh_pv;              // [B][sh_kc] (t_pv)
    int sh_kc;
    const float *sh_w1c, *sh_w1r;    // lz_model::sh_w1c / sh_w1r
    float *sh_part;                  // [B][3 heads: value, policy, value prefix][H/16 unit tiles][32 hidden]
    unsigned long long *stamp;       // optional [2] start / end stamps like lz_chain_args::stamp (k_lstm2 only)
};
void lz_launch_lstm(const lz_lstm_args &a, hipStream_t s);
// One launch per simulation (opt-in, LZ_SIM_ONE_LAUNCH=1; lz_nn.hip: k_sim_fused): the LSTM launch of simulation s - 1 and the tree-fused chain
// launch of simulation s as two phases of one launch whose 16-root groups hand the head partials over through one XCD's L2.  The control
// block (device memory, zeroed once per search): ticket counters per XCD, the fault word (a bounded spin ran out: the results are void),
// one arrival counter per (launch, group).
struct lz_res_ctl {
    unsigned xcc_count[16];
    unsigned fault;
    unsigned pad[15];
    unsigned flags[1];          // [launches][B / 16 groups]
};
struct lz_resident_args {
    lz_res_ctl *ctl;
    int launch, ngroups, B, BA;
};
size_t lz_fused_ctl_bytes(int B, int nlaunches);
bool lz_launch_sim_fused(const lz_lstm_args &la, const lz_chain_args &ca, const lz_tree_step &step, void *ctl, int launch, hipStream_t s);
// host: wcat [4H][K] (row 4*unit + gate) -> fragment order for lz_lstm_args::wf (4*H*K floats)
void lz_lstm_pack_fragments(const float *wcat, int H, int K, float *out);

// heads: Linear + BN + ReLU -> Linear [-> softmax.support -> h^-1].  Input element k of root b is read at
//   in[b * env_stride + (k / 16) * pix_stride + k % 16]   (k = pixel*16 + channel for the conv heads).
struct lz_head_desc {
    const float *in;
    int env_stride, pix_stride;
    const float *w1, *b1, *s1, *t1;  // Linear [HID][K1], bias, folded BN1d
    const float *w2t, *b2;           // second Linear as [HID / 4][NOUT][4] (coalesced float4 per output column), bias [NOUT]
    int K1, NOUT;
    int categorical;       // 1: softmax . support -> inverse scalar transform -> out_scalar[B]
    float support_min;     // support = support_min + k (step 1)
    float *out_logits;     // optional [B][NOUT]
    float *out_scalar;     // [B] (categorical)
    float *out_expect;     // optional [B] (categorical): softmax . support before the inverse scalar transform (parity tests)
};
void lz_launch_heads(const lz_head_desc *heads, int nheads, int B, int HID, hipStream_t s);
extern unsigned long long *lz_debug_heads_ts;   // null in production

// Dense layer for the vector-observation (MLP) model family, split over the chip in both dimensions:
//   out = epilogue( T(x) . W^T + bias ),   T = the PRODUCER's deferred LayerNorm / activation / residual, applied on load
// A workgroup owns 16 rows x 64 columns (4 waves x one 16-column MFMA tile), so a 256 x 256 layer at B = 256 is 64
// workgroups that each pull 64 KB of weights (a workgroup that owned whole rows would pull the entire matrix through one
// CU at ~10 B/clk: 13 us per layer, measured).  LayerNorm needs whole rows, so a layer with a LayerNorm stores its raw
// output and its consumers normalise while they stage their input; one consumer can also write the transformed rows
// back (the next latent state of the pool).  Up to 4 independent layers per launch (blockIdx.z).
struct lz_dense_job {
    const float *x;            // [B][K1] raw rows, or a pool base when x_gather != null (row b from slot x_gather[b])
    const int32_t *x_gather;
    int64_t x_slot_stride;
    int K1;
    const float *in_ln_g, *in_ln_b;  // deferred LayerNorm of the producer over the K1 columns (null: none)
    float in_ln_eps;
    int in_act;                // deferred activation of the producer: 0 none, 1 ReLU, 2 GELU(tanh)
    const float *in_res;       // residual rows [B][K1] added after the activation (a pool base with in_res_gather)
    const int32_t *in_res_gather;
    int64_t in_res_slot_stride;
    float *in_out;             // optional: the transformed rows, written by the column-group-0 workgroups ([B][K1])
    int in_minmax;             // 1: after the transforms above the row is renormalised to [0, 1] over its K1 columns -- (x - min) / max(max - min, 1e-8):
                               // state_norm=True of the MLP models (lzero/model/utils.py:242-271); what in_out receives is the renormalised row
    const float *x2;           // second input block (the action encoding); mode 1: float rows [B][K2]
    const int32_t *x2_idx;     // mode 2: one-hot(x2_idx[b]) of width K2 ; mode 3: the scalar x2_idx[b] / x2_div
    float x2_div;
    int K2, x2_mode;
    const float *wf;           // MFMA-fragment order [Np/16][Kp/16][64 lanes][4]: lane (n = l%16, g = l/16) holds W[n0+n][k0+4g..4g+3]
    const float *bias;         // [N]
    const float *scale, *shift;  // optional [N] folded eval-mode BatchNorm1d
    int N, act;                // act: applied here only when this layer has no LayerNorm (otherwise deferred to the consumers)
    int final;                 // 2: columns >= final_split: exp(clamp(v, -20, 2)), columns < final_split: tanh when final_tanh
    int final_split, final_tanh;
    float *out;                // [B][N]
};
struct lz_dense_args {
    lz_dense_job job[4];
    int njobs, B;
};
void lz_launch_dense(const lz_dense_args &a, hipStream_t s);

// row finisher: softmax(logits) . support -> InverseScalarTransform (scaling_transform.py:82-92), one wave per row
struct lz_rowfinal_job {
    const float *logits;       // [B][N]
    int N;
    float support_min;
    float *out_scalar;         // [B]
    int scalar;                // 1: categorical_distribution=False -- N = 1, the head's output IS the scaled scalar: h^-1(logits[b]) (scaling_transform.py:84-92)
};
struct lz_rowfinal_args {
    lz_rowfinal_job job[4];
    int njobs, B;
};
void lz_launch_rowfinal(const lz_rowfinal_args &a, hipStream_t s);

// Debugging / parity: out[i] = lz_inverse_scalar_transform(in[i]) (lz_hinv.h) evaluated by the copy of the function compiled into
// lz_nn.hip (which = 0: the head kernels and the split heads) or lz_dense.hip (which = 1: k_rowfinal).  Device pointers.
void lz_launch_hinv_nn(const float *d_in, float *d_out, int64_t n, hipStream_t s);
void lz_launch_hinv_dense(const float *d_in, float *d_out, int64_t n, hipStream_t s);
