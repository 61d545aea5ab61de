// lz_model.h -- device-side weight containers and the state_dict ingestion helpers shared by the convolutional
// (lz_search.hip) and the vector-observation / MLP (lz_mlp.hip) model families.
#pragma once
#include <math.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "lz_internal.h"
#include "lz_nn_kernels.h"

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
};

struct ConvW {
    float *w = nullptr, *scale = nullptr, *shift = nullptr;
    float *wf = nullptr;  // MFMA-fragment order [cout/16][9][cin/16][64 lanes][4] for the LDS-resident conv chain
    float *uf = nullptr;  // Winograd F(2x2, 3x3) transform U = G g G^T of the same weights, [cout/16][16][cin/16][64 lanes][4] (tower convs)
    float *uc = nullptr;  // the same transform in the order of the LDS-resident Winograd chain (k_chain_w): [16 points][cin/4][64 = cout][4]
    void *wb = nullptr;   // fast mode: bf16 fragments of the direct form for k_chain_b, [2 k halves][cout/16][9 taps][64 lanes][8 bf16]
    void *wt = nullptr;   // fast mode: bf16 fragments for the tower kernel k_conv_bf, [cout/16][9 taps x cin/32][64 lanes][8 bf16]
    void *w3 = nullptr;   // parity mode: the same fragments split exactly into three bf16 planes (hi | mid | lo) for k_conv_s3,
                          // [cout/16][9 taps x cin/32][3 planes][64 lanes][8 bf16]
    void *w3c = nullptr;  // parity mode, LDS-resident chains (k_chain_s3, k_chain_s3g): [2 k halves][cout/16][9 taps][3 planes][64 lanes][8 bf16]; w3cf = its fp32 values (refresh)
    float *w3cf = nullptr;
    float *w3f = nullptr; // the fp32 values of those fragments in the same order [cout/16][ks][64 lanes][8] (what a device-side refresh gathers; k_refresh_split3 turns it into w3)
    int cin = 0, cout = 0;
};
struct MlpW {
    std::vector<float> h_w1;   // host copy of w1 as uploaded ([32][K1], conv heads in (pixel, channel) order): lz_model_finalize builds the
                               // split-head fragments from it
    float *w1 = nullptr, *b1 = nullptr, *s1 = nullptr, *t1 = nullptr, *w2 = nullptr /* [32 / 4][NOUT][4], hidden width padded to the compiled 32 */, *b2 = nullptr;
    int K1 = 0, NOUT = 0;
};
// head MLP (Linear - BN1d - act - Linear) of the convolutional Sampled EfficientZero for the dense-layer kernels (k_dense, lz_dense.hip): any
// hidden width that is a multiple of 16 up to 256 (the reference's default), ReLU or GELU; weights in k_dense's fragment order
struct WideHead {
    float *w1f = nullptr, *b1 = nullptr, *s1 = nullptr, *t1 = nullptr, *w2f = nullptr, *b2 = nullptr;
    int K1 = 0, HID = 0, NOUT = 0;
};
struct C1W {
    float *w = nullptr, *b = nullptr, *s = nullptr, *t = nullptr;
};

// ------------------------------------------------------------------------------------------------
// Weight refresh on the device (lz_model_refresh_flat).  In the reference the collector searches with the learner's own nn.Module
// (lzero/policy/muzero.py:1049-1061): fresh weights cost nothing.  Here the weights live in kernel layouts, so a refresh re-lays them
// out -- lz_model_finalize does that on the host.  The SAME packer code (Builder below) is run once more in RECORDING mode on tensors
// that hold, instead of values, the CODE of their own elements: source space = [1.0f | the raw state_dict tensors, flat, in name
// order | derived tensors]; element i of the space has code (float)(i + 1), the literal 0.0f stays "zero" and the literal 1.0f is the
// code of the constant at index 0 -- so every pure re-ordering (fragment orders, transposes, column permutations, zero / one padding)
// runs unchanged and what it "uploads" is, read back as integers, the gather map of that device buffer.  The arithmetic packers
// (BatchNorm folding, Winograd U = G g G^T in binary64, the one-hot action table, the LSTM bias sum) register a DERIVED tensor computed
// by a kernel of its own (same operations, same order, contraction off: bit-identical to the host) and hand out its codes.  A refresh is
// then: the flat state_dict into the source buffer (one copy), four small kernels for the derived tensors, one gather kernel over
// all weight buffers -- on the engine's stream, no host synchronisation, same buffers (captured graphs stay valid).
struct RefreshRec {
    std::vector<std::string> names;          // raw tensors in flat order (std::map order == Python's sorted())
    std::vector<int64_t> offsets, sizes;     // in floats within the raw region
    int64_t raw_floats = 0, derived_floats = 0;
    struct Bn { int32_t w, b, mu, var, n, out; float eps; };
    struct Wino { int32_t w, cout, cin_total, cin, out; };          // out: U[cout][cin][16] (point = 4 i + j)
    struct Act { int32_t w, A, AE, C, SW, SH, out; };               // out: [A][SH * SW][C]
    struct Add { int32_t a, b, n, out; };
    struct Split3 { const float *src; void *dst; int64_t n; };      // a gathered fp32 fragment buffer -> its three bf16 planes (k_conv_s3), after the gather
    std::vector<Split3> split3;
    std::vector<Bn> bn;
    std::vector<Wino> wino;
    std::vector<Act> act;
    std::vector<Add> add;
    std::vector<int32_t> idx;                // concatenated gather maps (-1: 0.0f)
    struct Slot { float *dst; int64_t start, count; };
    std::vector<Slot> slots;
    bool ok = true;
    std::string why;
    int64_t derived(int64_t n) { const int64_t o = 1 + raw_floats + derived_floats; derived_floats += n; return o; }   // source index of a new derived region
    static float code(int64_t src_index) { return (float)(src_index + 1); }
    static int64_t index_of(float code) { return (int64_t)code - 1; }
    void fail(const std::string &w) { if (ok) { ok = false; why = w; } }
};
struct RefreshProgram {
    bool tried = false, usable = false;
    std::string why;
    std::vector<std::string> names;
    std::vector<int64_t> offsets, sizes;
    int64_t raw_floats = 0, src_floats = 0, out_floats = 0;
    float *d_src = nullptr;
    int32_t *d_idx = nullptr;
    void *d_slots = nullptr;    // RefreshRec::Slot[n_slots]
    int n_slots = 0;
    void *d_bn = nullptr, *d_wino = nullptr, *d_act = nullptr, *d_add = nullptr, *d_split3 = nullptr;
    int n_bn = 0, n_wino = 0, n_act = 0, n_add = 0, n_split3 = 0;
    int64_t split3_items = 0;
    int64_t wino_items = 0, act_items = 0;   // largest op, for the grid
    int bn_max = 0, add_max = 0;
    size_t n_allocs = 0;                     // the weight buffers the program was recorded for (lz_model::allocs.size())
    void *h_pin = nullptr;                   // pinned staging for a host-side flat state_dict
};

struct lz_model {
    lz_model_cfg cfg{};
    std::map<std::string, HostTensor> raw;
    bool raw_stale = false;              // a device refresh replaced the weights without passing through `raw`: lz_model_set_tensor
                                         // brings `raw` up to date from the source buffer first
    RefreshProgram refresh;
    bool finalized = false;
    std::vector<void *> allocs;          // device weight buffers in upload order
    std::vector<size_t> alloc_bytes;     // their sizes: a re-finalize with the same shapes copies in place (pointers stay valid,
    size_t alloc_cursor = 0;             // captured graphs stay valid); `realloc_happened` tells the engine to bump weights_gen
    bool realloc_happened = false;
    int HWl = 0;  // latent pixels (6x6 = 36 with downsample; obs_h*obs_w without)
    int GW = 6, GH = 6;
    ConvW rin;    // no-downsample input conv (weights [9][C][64] in rin.w)
    // representation
    float *first_w = nullptr, *first_s = nullptr, *first_t = nullptr;
    ConvW r1a, r1b, dn1, dn2, dn3, r2a, r2b, r3a, r3b;
    // num_res_blocks residual blocks each (two convolutions per block: [2 i] = conv1, [2 i + 1] = conv2)
    std::vector<ConvW> rep_res, dyn_res, pred_res;
    // dynamics
    ConvW dyn;
    float *act_table = nullptr;
    C1W rew_c;
    float *lstm_w = nullptr, *lstm_wf = nullptr, *lstm_b = nullptr, *vp_s = nullptr, *vp_t = nullptr;
    void *lstm_wb = nullptr;   // fast mode: the gate weights as bf16 fragments (k_lstm_b)
    MlpW fc_reward;
    // prediction
    C1W val_c, pol_c;
    MlpW fc_value, fc_policy;
    bool wide_heads = false;             // conv Sampled EfficientZero: the three heads run as dense layers + row finishers (lz_search.hip::wide_heads)
    WideHead wh_value, wh_policy, wh_reward;
    // split heads (EfficientZero, 6x6 latent; lz_search.hip): the first layers of the three head MLPs as MFMA B fragments for the
    // LSTM launch, sliced by LSTM unit tile u -- sh_w1c [H/16][9][4][64]: rows 36 u .. 36 u + 35 of the combined [value | policy]
    // head input (the 1x1-conv outputs t_pv, 1152 floats per root) x 64 hidden units (value 0..31 | policy 32..63, zero where the
    // input channel belongs to the other head); sh_w1r [H/16][4][2][64]: LSTM units 16 u .. 16 u + 15 x the value-prefix head's 32
    float *sh_w1c = nullptr, *sh_w1r = nullptr;
    // workspaces for initial inference
    int ws_B = 0;
    float *ws[3] = {nullptr, nullptr, nullptr};
    struct lz_mlp_model *mlp = nullptr;  // vector-observation family (model_type >= 2), see lz_mlp.hip
    int debug_stop = 0;  // lz_debug_set("stop_stage"): leave lz_initial_inference after stage k
};


namespace {

struct Builder {
    lz_model *m;
    std::string err;
    RefreshRec *rec = nullptr;   // recording mode: tensors hold codes, upload() records gather maps (see RefreshRec)
    // source index of element 0 of a (shadow) tensor
    int32_t src0(const HostTensor *t) const { return t && !t->data.empty() ? (int32_t)RefreshRec::index_of(t->data[0]) : -1; }
    const HostTensor *get(const std::string &name, std::initializer_list<int64_t> shape)
    {
        auto it = m->raw.find(name);
        if (it == m->raw.end()) { if (err.empty()) err = "missing tensor '" + name + "'"; return nullptr; }
        const HostTensor &t = it->second;
        if (t.shape.size() != shape.size() || !std::equal(shape.begin(), shape.end(), t.shape.begin())) {
            if (err.empty()) err = "tensor '" + name + "' has an unexpected shape";
            return nullptr;
        }
        return &t;
    }
    // The upload order of a finalize is a function of the model configuration and tensor shapes only, so a weight refresh
    // (lz_model_set_tensor + lz_model_finalize on a live engine, e.g. after a learner update) finds every buffer of the
    // previous finalize at the same position with the same size and overwrites it in place.
    float *upload(const std::vector<float> &v)
    {
        const size_t bytes = v.size() * 4;
        float *d = nullptr;
        if (rec) {   // the buffer of this position exists (same configuration, same shapes): record what it is gathered from
            if (m->alloc_cursor >= m->allocs.size() || m->alloc_bytes[m->alloc_cursor] != bytes) {
                rec->fail("weight buffer layout changed between the finalize and its recording");
                m->alloc_cursor++;
                return nullptr;
            }
            d = (float *)m->allocs[m->alloc_cursor++];
            rec->slots.push_back(RefreshRec::Slot{d, (int64_t)rec->idx.size(), (int64_t)v.size()});
            for (float c : v) {
                if (!(c >= 0.0f && c < 16777216.0f && c == floorf(c))) { rec->fail("a packer produced a value that is not an element code"); c = 0.0f; }
                rec->idx.push_back((int32_t)c - 1);
            }
            return d;
        }
        if (m->alloc_cursor < m->allocs.size() && m->alloc_bytes[m->alloc_cursor] == bytes) {
            d = (float *)m->allocs[m->alloc_cursor];
        } else {
            for (size_t i = m->alloc_cursor; i < m->allocs.size(); ++i) (void)hipFree(m->allocs[i]);  // shapes changed from here on
            m->allocs.resize(m->alloc_cursor);
            m->alloc_bytes.resize(m->alloc_cursor);
            if (lz_dev_malloc((void **)&d, bytes ? bytes : 4) != hipSuccess) { if (err.empty()) err = "hipMalloc failed for weights"; return nullptr; }
            m->allocs.push_back(d);
            m->alloc_bytes.push_back(bytes);
            m->realloc_happened = true;
        }
        m->alloc_cursor++;
        if (bytes && hipMemcpy(d, v.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) { if (err.empty()) err = "hipMemcpy failed for weights"; return nullptr; }
        return d;
    }
    // recording pass: the buffer at this upload position is produced by a kernel of its own (not by the gather): claim it, no map
    void *claim(size_t bytes_raw)
    {
        const size_t bytes = ((bytes_raw + 3) / 4) * 4;   // upload_u16 rounds up to whole floats
        if (m->alloc_cursor >= m->allocs.size() || m->alloc_bytes[m->alloc_cursor] != bytes) {
            rec->fail("weight buffer layout changed between the finalize and its recording");
            m->alloc_cursor++;
            return nullptr;
        }
        return m->allocs[m->alloc_cursor++];
    }
    // eval-mode BatchNorm -> y = x*scale + shift
    void bn(const std::string &prefix, int n, std::vector<float> &scale, std::vector<float> &shift)
    {
        const HostTensor *w = get(prefix + ".weight", {n}), *b = get(prefix + ".bias", {n}),
                         *mu = get(prefix + ".running_mean", {n}), *var = get(prefix + ".running_var", {n});
        scale.assign(n, 1.0f);
        shift.assign(n, 0.0f);
        if (!w || !b || !mu || !var) return;
        if (rec) {   // derived tensor [scale n | shift n], computed by k_refresh_bn with the expressions below
            const int64_t o = rec->derived(2 * (int64_t)n);
            rec->bn.push_back(RefreshRec::Bn{src0(w), src0(b), src0(mu), src0(var), n, (int32_t)o, m->cfg.bn_eps});
            for (int i = 0; i < n; ++i) { scale[i] = RefreshRec::code(o + i); shift[i] = RefreshRec::code(o + n + i); }
            return;
        }
        for (int i = 0; i < n; ++i) {
            const float inv = 1.0f / sqrtf(var->data[i] + m->cfg.bn_eps);
            scale[i] = w->data[i] * inv;
            shift[i] = b->data[i] - mu->data[i] * scale[i];
        }
    }
    // conv weight [cout][cin_total][3][3] -> packed [cout/16][9][16][cin] (first `cin` input channels)
    ConvW conv(const std::string &wname, const std::string &bnprefix, int cout, int cin_total, int cin)
    {
        ConvW c;
        c.cin = cin;
        c.cout = cout;
        const HostTensor *w = get(wname, {cout, cin_total, 3, 3});
        std::vector<float> sc(cout, 1.0f), sh(cout, 0.0f);
        if (!bnprefix.empty()) bn(bnprefix, cout, sc, sh);
        if (!w) return c;
        std::vector<float> p((size_t)cout * 9 * cin);
        for (int co = 0; co < cout; ++co)
            for (int t = 0; t < 9; ++t)
                for (int ci = 0; ci < cin; ++ci)
                    p[(((size_t)(co / 16) * 9 + t) * 16 + co % 16) * cin + ci] = w->data[(((size_t)co * cin_total + ci) * 9) + t];
        c.w = upload(p);
        c.scale = upload(sc);
        c.shift = upload(sh);
        {
            std::vector<float> f((size_t)cout * 9 * cin);
            for (int nt = 0; nt < cout / 16; ++nt)
                for (int t = 0; t < 9; ++t)
                    for (int g = 0; g < cin / 16; ++g)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 4; ++j) {
                                const int n = lane & 15, kq = lane >> 4, ci = g * 16 + kq * 4 + j, co = nt * 16 + n;
                                f[((((size_t)nt * 9 + t) * (cin / 16) + g) * 64 + lane) * 4 + j] = w->data[(((size_t)co * cin_total + ci) * 9) + t];
                            }
            c.wf = upload(f);
        }
        return c;
    }
    ConvW resconv(const std::string &prefix, int idx, int cout, int cin, bool winograd = false, bool wchain = false, bool s3chain = false)  // ding ResBlock convN = Sequential(conv, bn[, act])
    {
        const std::string p = prefix + ".conv" + std::to_string(idx);
        ConvW c = conv(p + ".0.weight", p + ".1", cout, cin, cin);
        if (winograd) c.uf = wino(p + ".0.weight", cout, cin);
        if (wchain) c.uc = wino_chain(p + ".0.weight", cout, cin, cin);
        if (wchain && m->cfg.precision == 1) c.wb = bf16_chain(p + ".0.weight", cout, cin, cin);
        if (s3chain) split3_chain(p + ".0.weight", cout, cin, cin, c);   // parity mode: k_chain_s3 (6x6) / k_chain_s3g (8x8, 9x9, 6x7, 4x4)
        return c;
    }
    // fp32 -> bf16, round to nearest even (what v_cvt_pk_bf16_f32 does to the activations on the device)
    static uint16_t bf16_rne(float f)
    {
        uint32_t u;
        memcpy(&u, &f, 4);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    }
    float *upload_u16(const std::vector<uint16_t> &v)
    {
        std::vector<float> f((v.size() + 1) / 2, 0.0f);
        memcpy(f.data(), v.data(), v.size() * 2);
        return upload(f);
    }
    // fast mode, k_conv_bf: lane (n = l & 15, kq = l >> 4) of k step (tap t, 32-channel block kc) holds W[co = 16 nt + n][ci = 32 kc + 8 kq + j][t]
    void *bf16_tower(const std::string &wname, int cout, int cin)
    {
        const HostTensor *w = get(wname, {cout, cin, 3, 3});
        if (!w || (cin & 31) || (cout & 15)) return nullptr;
        const int KC = cin / 32, KS = 9 * KC;
        std::vector<uint16_t> f((size_t)cout * cin * 9);
        for (int nt = 0; nt < cout / 16; ++nt)
            for (int t = 0; t < 9; ++t)
                for (int kc = 0; kc < KC; ++kc)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int co = nt * 16 + (lane & 15), ci = kc * 32 + (lane >> 4) * 8 + j;
                            f[((((size_t)nt * KS + t * KC + kc) * 64) + lane) * 8 + j] = bf16_rne(w->data[((size_t)co * cin + ci) * 9 + t]);
                        }
        return upload_u16(f);
    }
    // parity mode, k_conv_s3: the k_conv_bf fragment order with every weight split exactly into three bf16 terms (hi = rne(w), mid =
    // rne(w - hi), lo = rne(w - hi - mid)); lane (n = l & 15, kq = l >> 4) of k step (tap t, 32-channel block kc) holds
    // W[co = 16 nt + n][ci = 32 kc + 8 kq + j][t], j = 0..7, three planes per k step.  The fp32 values in the same order are kept on the
    // device too (w3f): a device-side refresh gathers them and k_refresh_split3 rebuilds the planes.
    static void split3(float w, uint16_t &h, uint16_t &m, uint16_t &l)
    {
        auto tof = [](uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; };
        h = bf16_rne(w);
        const float r1 = w - tof(h);
        m = bf16_rne(r1);
        const float r2 = r1 - tof(m);
        l = bf16_rne(r2);
    }
    void split3_tower(const std::string &wname, int cout, int cin, ConvW &c)
    {
        const HostTensor *w = get(wname, {cout, cin, 3, 3});
        if (!w || (cin & 31) || (cout & 15)) return;
        const int KC = cin / 32, KS = 9 * KC;
        std::vector<float> f((size_t)cout * cin * 9);
        for (int nt = 0; nt < cout / 16; ++nt)
            for (int t = 0; t < 9; ++t)
                for (int kc = 0; kc < KC; ++kc)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int co = nt * 16 + (lane & 15), ci = kc * 32 + (lane >> 4) * 8 + j;
                            f[((((size_t)nt * KS + t * KC + kc) * 64) + lane) * 8 + j] = w->data[((size_t)co * cin + ci) * 9 + t];
                        }
        c.w3f = upload(f);
        // the three planes: values on the host path; on the recording pass only the buffer is claimed (k_refresh_split3 fills it from w3f)
        std::vector<uint16_t> p3(f.size() * 3);
        if (!rec) {
            const size_t per = (size_t)64 * 8;
            for (size_t blk = 0; blk < f.size() / per; ++blk)
                for (size_t e = 0; e < per; ++e) {
                    uint16_t h, m, l;
                    split3(f[blk * per + e], h, m, l);
                    p3[(blk * 3 + 0) * per + e] = h; p3[(blk * 3 + 1) * per + e] = m; p3[(blk * 3 + 2) * per + e] = l;
                }
            c.w3 = upload_u16(p3);
        } else {
            c.w3 = claim(p3.size() * 2);
            if (c.w3 && c.w3f) rec->split3.push_back(RefreshRec::Split3{c.w3f, c.w3, (int64_t)f.size()});
        }
    }
    // parity mode, k_chain_s3: k_chain_b's fragment order ([kh][nt][tap][64 lanes][8]: lane (n = l & 15, kq = l >> 4) holds
    // W[co = 16 nt + n][ci = 32 kh + 8 kq + j][t]) with every weight split into three bf16 planes per (kh, nt, tap) block
    void split3_chain(const std::string &wname, int cout, int cin_total, int cin, ConvW &c)
    {
        const HostTensor *w = get(wname, {cout, cin_total, 3, 3});
        if (!w || cout != 64 || cin != 64) return;
        std::vector<float> f((size_t)2 * 4 * 9 * 64 * 8);
        for (int kh = 0; kh < 2; ++kh)
            for (int nt = 0; nt < 4; ++nt)
                for (int t = 0; t < 9; ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int co = nt * 16 + (lane & 15), ci = kh * 32 + (lane >> 4) * 8 + j;
                            f[((((size_t)kh * 4 + nt) * 9 + t) * 64 + lane) * 8 + j] = w->data[((size_t)co * cin_total + ci) * 9 + t];
                        }
        c.w3cf = upload(f);
        std::vector<uint16_t> p3(f.size() * 3);
        if (!rec) {
            const size_t per = (size_t)64 * 8;
            for (size_t blk = 0; blk < f.size() / per; ++blk)
                for (size_t e = 0; e < per; ++e) {
                    uint16_t h, m2, l;
                    split3(f[blk * per + e], h, m2, l);
                    p3[(blk * 3 + 0) * per + e] = h; p3[(blk * 3 + 1) * per + e] = m2; p3[(blk * 3 + 2) * per + e] = l;
                }
            c.w3c = upload_u16(p3);
        } else {
            c.w3c = claim(p3.size() * 2);
            if (c.w3c && c.w3cf) rec->split3.push_back(RefreshRec::Split3{c.w3cf, c.w3c, (int64_t)f.size()});
        }
    }
    // fast mode, k_chain_b: wave (nt, kh) multiplies the 16-channel output tile nt by input channels 32 kh .. + 31 of tap t; lane (n = l & 15,
    // kq = l >> 4) holds W[co = 16 nt + n][ci = 32 kh + 8 kq + j][t], j = 0..7 -- the B operand of v_mfma_f32_16x16x32_bf16
    void *bf16_chain(const std::string &wname, int cout, int cin_total, int cin)
    {
        const HostTensor *w = get(wname, {cout, cin_total, 3, 3});
        if (!w || cout != 64 || cin != 64) return nullptr;
        std::vector<uint16_t> f((size_t)2 * 4 * 9 * 64 * 8);
        for (int kh = 0; kh < 2; ++kh)
            for (int nt = 0; nt < 4; ++nt)
                for (int t = 0; t < 9; ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int co = nt * 16 + (lane & 15), ci = kh * 32 + (lane >> 4) * 8 + j;
                            f[((((size_t)kh * 4 + nt) * 9 + t) * 64 + lane) * 8 + j] = bf16_rne(w->data[((size_t)co * cin_total + ci) * 9 + t]);
                        }
        return upload_u16(f);
    }
    static void wino_u(const float *g, double (&U)[4][4])  // U = G g G^T, binary64
    {
        static const double Gm[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
        double t[4][3];
        for (int i = 0; i < 4; ++i)
            for (int k = 0; k < 3; ++k) t[i][k] = Gm[i][0] * g[0 * 3 + k] + Gm[i][1] * g[1 * 3 + k] + Gm[i][2] * g[2 * 3 + k];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) U[i][j] = t[i][0] * Gm[j][0] + t[i][1] * Gm[j][1] + t[i][2] * Gm[j][2];
    }
    // order of k_chain_w: wave w streams the points 4w .. 4w+3; lane = output channel; one float4 = four consecutive input channels
    // (the first `cin` of cin_total: the dynamics conv's one-hot action planes go through the action table)
    float *wino_chain(const std::string &wname, int cout, int cin_total, int cin)
    {
        const HostTensor *w = get(wname, {cout, cin_total, 3, 3});
        if (!w || cout != 64 || (cin & 3)) return nullptr;
        std::vector<float> f((size_t)16 * cin * cout);
        const int64_t o = rec ? rec->derived((int64_t)cout * cin * 16) : 0;   // derived tensor U[cout][cin][16] (k_refresh_wino)
        if (rec) rec->wino.push_back(RefreshRec::Wino{src0(w), cout, cin_total, cin, (int32_t)o});
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci) {
                double U[4][4];
                if (!rec) wino_u(&w->data[((size_t)co * cin_total + ci) * 9], U);
                for (int p = 0; p < 16; ++p)
                    f[((((size_t)p * (cin / 4) + ci / 4) * cout) + co) * 4 + (ci & 3)] =
                        rec ? RefreshRec::code(o + ((int64_t)co * cin + ci) * 16 + p) : (float)U[p / 4][p % 4];
            }
        return upload(f);
    }
    // U = G g G^T per (cout, cin) filter, G = [[1,0,0],[1/2,1/2,1/2],[1/2,-1/2,1/2],[0,0,1]] (Lavin & Gray 2016), computed in
    // binary64 and rounded once; fragment order of k_conv_wino: lane (n = l & 15, kq = l >> 4) holds U_p[ci = 16 g + 4 kq + j][co = 16 nt + n]
    float *wino(const std::string &wname, int cout, int cin)
    {
        const HostTensor *w = get(wname, {cout, cin, 3, 3});
        if (!w) return nullptr;
        const int G = cin / 16;
        std::vector<float> f((size_t)cout * cin * 16);
        const int64_t o = rec ? rec->derived((int64_t)cout * cin * 16) : 0;
        if (rec) rec->wino.push_back(RefreshRec::Wino{src0(w), cout, cin, cin, (int32_t)o});
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci) {
                double U[4][4];
                if (!rec) wino_u(&w->data[((size_t)co * cin + ci) * 9], U);
                const int nt = co / 16, n = co % 16, gq = ci / 16, kq = (ci % 16) / 4, jj = ci % 4, lane = kq * 16 + n;
                for (int p = 0; p < 16; ++p)
                    f[((((size_t)nt * 16 + p) * G + gq) * 64 + lane) * 4 + jj] =
                        rec ? RefreshRec::code(o + ((int64_t)co * cin + ci) * 16 + p) : (float)U[p / 4][p % 4];
            }
        return upload(f);
    }
    C1W conv1x1(const std::string &cprefix, const std::string &bnprefix, int cout, int cin)
    {
        C1W c;
        const HostTensor *w = get(cprefix + ".weight", {cout, cin, 1, 1}), *b = get(cprefix + ".bias", {cout});
        std::vector<float> sc, sh;
        bn(bnprefix, cout, sc, sh);
        if (!w || !b) return c;
        c.w = upload(w->data);
        c.b = upload(b->data);
        c.s = upload(sc);
        c.t = upload(sh);
        return c;
    }
    // [N][K] row-major -> k_dense's MFMA-fragment order [Np/16][Kp/16][64 lanes][4] (lane (n = l & 15, g = l >> 4) holds W[n0 + n][k0 + 4 g .. + 3])
    float *pack_dense(const std::vector<float> &w, int N, int K)
    {
        const int Np = (N + 15) & ~15, Kp = (K + 15) & ~15, KB = Kp / 16;
        std::vector<float> f((size_t)Np * Kp, 0.0f);
        for (int nt = 0; nt < Np / 16; ++nt)
            for (int kb = 0; kb < KB; ++kb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int q = 0; q < 4; ++q) {
                        const int n = nt * 16 + (lane & 15), k = kb * 16 + (lane >> 4) * 4 + q;
                        if (n < N && k < K) f[(((size_t)nt * KB + kb) * 64 + lane) * 4 + q] = w[(size_t)n * K + k];
                    }
        return upload(f);
    }
    // the same head for the dense-layer kernels (conv Sampled EfficientZero); conv_flat as in mlp() below
    WideHead wide_head(const std::string &prefix, int K1, int HID, int NOUT, bool conv_flat, int HC, int HW)
    {
        WideHead o;
        o.K1 = K1; o.HID = HID; o.NOUT = NOUT;
        const HostTensor *w1 = get(prefix + ".0.weight", {HID, K1}), *b1 = get(prefix + ".0.bias", {HID}),
                         *w2 = get(prefix + ".3.weight", {NOUT, HID}), *b2 = get(prefix + ".3.bias", {NOUT});
        std::vector<float> sc, sh;
        bn(prefix + ".1", HID, sc, sh);
        if (!w1 || !b1 || !w2 || !b2) return o;
        std::vector<float> w1p(w1->data);
        if (conv_flat) {
            for (int u = 0; u < HID; ++u)
                for (int p = 0; p < HW; ++p)
                    for (int c = 0; c < HC; ++c) w1p[(size_t)u * K1 + p * HC + c] = w1->data[(size_t)u * K1 + c * HW + p];
        }
        o.w1f = pack_dense(w1p, HID, K1);
        o.b1 = upload(b1->data);
        o.s1 = upload(sc);
        o.t1 = upload(sh);
        o.w2f = pack_dense(w2->data, NOUT, HID);
        o.b2 = upload(b2->data);
        return o;
    }
    // Linear - BN1d - ReLU - Linear; conv_flat: K1 = HC*HW in the reference's (channel, pixel) order ->
    // permute the columns to this engine's (pixel, channel) order
    // The head kernel is compiled for 32 hidden units; a narrower head (tictactoe: [8]) is padded with units whose weights,
    // bias and BatchNorm shift are zero: they output relu(0) = 0 and add exact zeros in the second layer, so the real units'
    // sums keep their order and their bits.
    MlpW mlp(const std::string &prefix, int K1, int HID, int NOUT, bool conv_flat, int HC, int HW)
    {
        constexpr int HP = 32;
        MlpW o;
        o.K1 = K1;
        o.NOUT = NOUT;
        const HostTensor *w1 = get(prefix + ".0.weight", {HID, K1}), *b1 = get(prefix + ".0.bias", {HID}),
                         *w2 = get(prefix + ".3.weight", {NOUT, HID}), *b2 = get(prefix + ".3.bias", {NOUT});
        std::vector<float> sc, sh;
        bn(prefix + ".1", HID, sc, sh);
        if (!w1 || !b1 || !w2 || !b2) return o;
        if (HID > HP) { if (err.empty()) err = "head hidden width above 32"; return o; }
        std::vector<float> w1p((size_t)HP * K1, 0.0f), b1p(HP, 0.0f), scp(HP, 1.0f), shp(HP, 0.0f);
        for (int u = 0; u < HID; ++u) {
            b1p[u] = b1->data[u]; scp[u] = sc[u]; shp[u] = sh[u];
            for (int k = 0; k < K1; ++k) w1p[(size_t)u * K1 + k] = w1->data[(size_t)u * K1 + k];
        }
        if (conv_flat) {
            for (int u = 0; u < HID; ++u)
                for (int p = 0; p < HW; ++p)
                    for (int c = 0; c < HC; ++c) w1p[(size_t)u * K1 + p * HC + c] = w1->data[(size_t)u * K1 + c * HW + p];
        }
        o.w1 = upload(w1p);
        o.h_w1 = w1p;
        o.b1 = upload(b1p);
        o.s1 = upload(scp);
        o.t1 = upload(shp);
        // second layer as [32 / 4][NOUT][4]: thread n takes its column's weights with eight coalesced 16-byte loads (a wave reads
        // 1 KB per instruction) instead of thirty-two 4-byte ones from a plain transpose (measured: no change of the 8.2 us
        // launch, which is a chain of dependent stages, not load-issue bound; kept for the fewer instructions)
        std::vector<float> w2n((size_t)NOUT * HP, 0.0f);
        for (int n = 0; n < NOUT; ++n)
            for (int k = 0; k < HID; ++k) w2n[((size_t)(k / 4) * NOUT + n) * 4 + k % 4] = w2->data[(size_t)n * HID + k];
        o.w2 = upload(w2n);
        o.b2 = upload(b2->data);
        return o;
    }
};


}  // namespace
