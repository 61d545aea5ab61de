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

// first layer of DownSample: conv3x3 stride 2 from NCHW obs [B][C][H][W] (C <= 4... any small C) to NHWC
void lz_launch_conv_first(const float *obs_nchw, const float *w /*[9][C][Cout]*/, const float *scale,
                          const float *shift, float *out, int B, int C, int H, int W, int Cout, hipStream_t s);

// AvgPool2d(kernel 3, stride 2, pad 1, count_include_pad) on NHWC
void lz_launch_avgpool(const float *in, float *out, int B, int Hin, int Win, int C, hipStream_t s);

// conv1x1 (CIN -> Cout<=32) + BN + ReLU on NHWC [B][HW][CIN] -> [B][HW][Cout]
void lz_launch_conv1x1(const float *in, const float *w /*[Cout][CIN]*/, const float *bias, const float *scale,
                       const float *shift, float *out, int B, int HW, int CIN, int Cout, hipStream_t s);

// one LSTM step (nn.LSTM, 1 layer) fused with BatchNorm1d + ReLU of the output:
//   gates = [x | h] . Wcat^T + bias ; c' = sig(f) c + sig(i) tanh(g) ; h' = sig(o) tanh(c')
struct lz_lstm_args {
    const float *x;          // [B][KX]
    const float *h_pool, *c_pool;  // pools [NN][B][H]; row b read from slot gather_ix[b]
    const int32_t *gather_ix;      // [B]
    const float *wcat;       // [4H][KX+H], row n = 4*unit + gate (gate order i,f,g,o)
    const float *bias;       // [4H] same order (b_ih + b_hh)
    const float *bn_scale, *bn_shift;  // [H]
    const int32_t *search_len;  // [B] (reset when search_len % horizon == 0); may be null => no reset
    int horizon;
    float *h_out, *c_out;    // [B][H] destination slot of the pools
    float *hbn_out;          // [B][H] relu(bn(h'))
    int B, KX, H;
};
void lz_launch_lstm(const lz_lstm_args &a, hipStream_t s);

// prediction / reward heads: (conv1x1 + BN + ReLU) -> Linear + BN + ReLU -> Linear [-> softmax.support -> h^-1]
struct lz_head_desc {
    const float *in;       // conv head: NHWC latent [B][HW][C];  vector head: [B][K1]
    int has_conv;          // 1: conv1x1 C -> HC first
    const float *cw, *cb, *cscale, *cshift;  // conv1x1 [HC][C], bias, folded BN
    const float *w1, *b1, *s1, *t1;  // Linear [HID][K1] (K1 index = pixel*HC + channel), bias, folded BN1d
    const float *w2, *b2;            // Linear [NOUT][HID]
    int K1, NOUT;
    int categorical;       // 1: softmax . support -> inverse scalar transform -> out_scalar[B]
    float support_min;     // support = support_min + k (step 1)
    float *out_logits;     // optional [B][NOUT]
    float *out_scalar;     // [B] (categorical)
};
void lz_launch_heads(const lz_head_desc *heads, int nheads, int B, int HW, int C, int HC, int HID, hipStream_t s);
