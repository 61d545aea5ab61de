// lz_search.hip -- weight ingestion (reference state_dict names -> kernel layouts), the fused
// initial_inference / recurrent_inference launch chains and the on-device search loop.
//
// Reference call chain replaced (LightZero v0.2.0):
//   EfficientZeroPolicy._forward_collect         lzero/policy/efficientzero.py:572-615
//   EfficientZeroMCTSCtree.search                lzero/mcts/tree_search/mcts_ctree.py:745-876
//   EfficientZeroModel.initial/recurrent_inference  lzero/model/efficientzero_model.py:203-273
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "lz_internal.h"
#include "lz_nn_kernels.h"

#include "lz_model.h"
#include "lz_mlp.h"

static void refresh_program_free(lz_model *m);
void lz_model_destroy(lz_model *m)
{
    if (!m) return;
    refresh_program_free(m);
    for (void *p : m->allocs) (void)hipFree(p);
    for (int i = 0; i < 3; ++i) if (m->ws[i]) (void)hipFree(m->ws[i]);
    if (m->mlp) lz_mlp_model_destroy(m->mlp);
    delete m;
}

// An engine holds one model; creating another one replaces it.  model_uid tells every roots handle and every host-side model
// object that what they were sized for / bound to is gone (lz_engine_model_uid; the pools are re-allocated, stale users fail).
static int replace_model(lz_engine *e)
{
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    if (e->model) lz_model_destroy(e->model);
    e->model = new (std::nothrow) lz_model();
    if (!e->model) { lz_set_error("out of host memory"); return LZ_ERR_NOMEM; }
    e->model_uid++;
    e->weights_gen++;
    return LZ_OK;
}

extern "C" uint64_t lz_engine_model_uid(lz_engine *e) { return e ? e->model_uid : 0; }

extern "C" int lz_model_create(lz_engine *e, const lz_model_cfg *cfg)
{
    LZ_REQUIRE(e != nullptr && cfg != nullptr, "NULL argument");
    LZ_REQUIRE(cfg->model_type >= 0 && cfg->model_type <= 4, "model_type must be 0 (EfficientZeroModel), 1 (MuZeroModel), 2 (MuZeroModelMLP), 3 (EfficientZeroModelMLP) or 4 (SampledEfficientZeroModelMLP)");
    LZ_REQUIRE(cfg->support_size > 0 && cfg->support_size <= 768, "support_size must be in [1, 768]");
    LZ_REQUIRE(cfg->reward_support_size >= 0 && cfg->reward_support_size <= 768, "reward_support_size must be in [0, 768] (0 = the value support)");
    LZ_REQUIRE(cfg->precision == 0 || cfg->precision == 2 || (cfg->precision == 1 && (cfg->model_type == 0 || cfg->model_type == 1) && cfg->num_of_sampled_actions == 0 &&
                                        cfg->downsample && (cfg->obs_h == 96 || cfg->obs_h == 64) && cfg->obs_c == 4 && cfg->num_channels == 64 &&
                                        (cfg->model_type == 1 || cfg->lstm_hidden_size == 512)),
               "precision must be 0 (fp32, parity mode), 2 (parity mode on the fp32 matrix instructions only) or 1 (bf16 fast mode: EfficientZeroModel (LSTM 512) / MuZeroModel with 4x96x96 -> 6x6x64 or 4x64x64 -> 8x8x64)");
    LZ_REQUIRE((cfg->state_norm == 0 && cfg->scalar_heads == 0) || cfg->model_type >= 2, "state_norm / scalar_heads (categorical_distribution=False): the MLP model family only");
    LZ_REQUIRE(cfg->scalar_heads == 0 || (cfg->support_size == 1 && (cfg->reward_support_size == 0 || cfg->reward_support_size == 1)), "scalar_heads: the heads have one output (support_size = 1)");
    LZ_REQUIRE(cfg->state_norm == 0 || cfg->num_channels % 4 == 0, "state_norm: latent_state_dim must be a multiple of 4");
    LZ_REQUIRE(cfg->reward_support_size == 0 || cfg->model_type == 1 || cfg->model_type == 2 || (cfg->reward_support_size == cfg->support_size && cfg->reward_support_min == cfg->support_min),
               "a reward support of its own: the MuZero models only (the EfficientZero drivers transform the value prefix with the VALUE handle, mcts_ctree.py:839-841)");
    if (cfg->model_type >= 2) {
        // vector observations: obs_c = observation_shape, num_channels = latent_state_dim; the layer widths come from the tensors
        LZ_REQUIRE(cfg->obs_c >= 1 && cfg->obs_h == 1 && cfg->obs_w == 1, "MLP models take obs_c = observation_shape, obs_h = obs_w = 1");
        LZ_REQUIRE(cfg->num_channels >= 16 && cfg->num_channels <= 1024, "latent_state_dim must be in [16, 1024]");
        LZ_REQUIRE(cfg->action_space_size > 0 && cfg->action_space_size <= 256, "action_space_size must be in [1, 256]");
        LZ_REQUIRE(cfg->action_encoding >= 0 && cfg->action_encoding <= 2, "action_encoding must be 0 (one_hot), 1 (not_one_hot) or 2 (continuous)");
        LZ_REQUIRE(cfg->action_encoding != 2 || cfg->model_type == 4, "continuous actions need the SampledEfficientZeroModelMLP");
        if (cfg->model_type != 2) {
            const int nchunk = (cfg->num_channels + cfg->lstm_hidden_size) / 64;
            LZ_REQUIRE(cfg->num_channels % 64 == 0 && cfg->lstm_hidden_size % 64 == 0 && (nchunk == 4 || nchunk == 6 || nchunk == 8 || nchunk == 12 || nchunk == 9 || nchunk == 13 || nchunk == 17),
                       "compiled LSTM shapes: (latent_state_dim + lstm_hidden_size) / 64 in {4, 6, 8, 9, 12, 13, 17}, both multiples of 64");
        }
        if (int rc = replace_model(e)) return rc;
        e->model->cfg = *cfg;
        if (e->model->cfg.bn_eps <= 0) e->model->cfg.bn_eps = 1e-5f;
        if (e->model->cfg.ln_eps <= 0) e->model->cfg.ln_eps = 1e-5f;
        e->model->GW = e->model->GH = e->model->HWl = 1;
        return LZ_OK;
    }
    LZ_REQUIRE(cfg->num_channels == 64 || cfg->num_channels == 32 || cfg->num_channels == 16, "num_channels must be 64, 32 or 16");
    LZ_REQUIRE(cfg->num_res_blocks >= 0 && cfg->num_res_blocks <= 3, "num_res_blocks must be 1, 2 or 3");
    if (cfg->num_channels != 64) {
        // the narrow chain (k_chain_small): the reference's small board-game models (gomoku 32 channels, tictactoe 16)
        LZ_REQUIRE(!cfg->downsample && (cfg->model_type == 1 || cfg->model_type == 0), "num_channels 32 / 16: models without downsample (board games)");
        LZ_REQUIRE(lz_chain_small_supported(cfg->obs_w, cfg->obs_h, cfg->num_channels), "no narrow-chain instance for this board: 3x3, 6x6, 6x7, 9x9");
    }
    if (cfg->downsample) {
        // the two sizes the reference models define a latent size for (efficientzero_model.py:121-124): 96x96 ends on a 6x6
        // latent; 64x64 (the shipped Atari configs, zoo/atari/config/atari_efficientzero_config.py:29) skips DownSample's
        // last pooling (common.py:355-359) and ends on 8x8
        LZ_REQUIRE(cfg->obs_h == cfg->obs_w && (cfg->obs_h == 96 || cfg->obs_h == 64),
                   "observation must be 96x96 or 64x64 on the downsample path (efficientzero_model.py:121-124)");
        LZ_REQUIRE(cfg->obs_c == 1 || cfg->obs_c == 3 || cfg->obs_c == 4 || cfg->obs_c == 12, "obs_c must be 1, 3, 4 or 12");
    } else {
        LZ_REQUIRE((cfg->obs_h == 9 && cfg->obs_w == 9) || (cfg->obs_h == 6 && cfg->obs_w == 7) || (cfg->obs_h == 6 && cfg->obs_w == 6) ||
                       (cfg->obs_h == 8 && cfg->obs_w == 8 && cfg->num_channels == 64) || (cfg->obs_h == 4 && cfg->obs_w == 4 && cfg->num_channels == 64) || (cfg->obs_h == 3 && cfg->obs_w == 3 && cfg->num_channels != 64),
                   "without downsample the compiled latent grids are 9x9 (Go), 8x8 and 4x4 (2048) with 64 channels, 6x7 (Connect4), 6x6 (gomoku) and, for the narrow models, 3x3 (tictactoe)");
        LZ_REQUIRE(cfg->obs_c >= 1 && cfg->obs_c <= 64, "obs_c must be in [1, 64]");
    }
    const bool conv_sampled = cfg->model_type == 0 && cfg->num_of_sampled_actions > 0;
    LZ_REQUIRE(cfg->num_of_sampled_actions == 0 || (conv_sampled && cfg->num_of_sampled_actions <= 64 && cfg->downsample && cfg->obs_c == 4 && cfg->num_channels == 64),
               "num_of_sampled_actions on a conv model = SampledEfficientZeroModel (conv): EfficientZero network + sampled tree, discrete actions, K in [1, 64], 4-channel downsampled observations");
    LZ_REQUIRE(cfg->activation == 0 || conv_sampled, "conv models: GELU (activation 1) exists for the convolutional Sampled EfficientZero only");
    // the value-prefix LSTM applies GELU behind its BatchNorm in two instances only (6x6 and 8x8 latents, hidden 512): any other shape would
    // silently compute relu(bn(h')) where the reference's DynamicsNetwork computes GELU
    LZ_REQUIRE(!(conv_sampled && cfg->activation == 1) || cfg->lstm_hidden_size == 512,
               "conv Sampled EfficientZero with GELU: lstm_hidden_size must be 512 (the value-prefix LSTM's GELU instances)");
    LZ_REQUIRE(cfg->head_channels == 16 && cfg->head_hidden >= 1 && (cfg->head_hidden <= 32 || (conv_sampled && cfg->head_hidden <= 256 && cfg->head_hidden % 16 == 0)),
               "head_channels must be 16 and head_hidden at most 32 (conv Sampled EfficientZero: a multiple of 16 up to 256)");
    LZ_REQUIRE(cfg->model_type == 1 || (cfg->lstm_hidden_size % 64 == 0 && cfg->lstm_hidden_size > 0), "lstm_hidden_size must be a multiple of 64");
    LZ_REQUIRE(cfg->action_space_size > 0 && cfg->action_space_size <= 256, "action_space_size must be in [1, 256]");
    LZ_REQUIRE(cfg->action_encoding == 0 || cfg->action_encoding == 1, "conv models: action_encoding 0 (one_hot) or 1 (not_one_hot)");
    if (cfg->model_type == 0) {  // the value-prefix LSTM reads [16 channels x latent pixels | hidden]: compiled K shapes
        const int gpix = cfg->downsample ? (cfg->obs_h == 64 ? 64 : 36) : cfg->obs_h * cfg->obs_w;
        const int K = cfg->head_channels * gpix + cfg->lstm_hidden_size;
        const bool frag = K % 16 == 0 && (K / 16 == 68 || K / 16 == 96 || K / 16 == 41 || K / 16 == 74 || K / 16 == 113), chunked = K % 64 == 0 && (K / 64 == 17 || K / 64 == 13 || K / 64 == 12 || K / 64 == 9);
        LZ_REQUIRE(frag || chunked, "no LSTM kernel instance for this (latent grid, lstm_hidden_size): hidden 512 on 3x3, 4x4, 6x6, 6x7, 8x8, 9x9 latents; hidden 256 on 6x6");
    }
    if (int rc = replace_model(e)) return rc;
    e->model->cfg = *cfg;
    if (e->model->cfg.bn_eps <= 0) e->model->cfg.bn_eps = 1e-5f;
    e->model->GW = cfg->downsample ? (cfg->obs_w == 64 ? 8 : 6) : cfg->obs_w;
    e->model->GH = cfg->downsample ? (cfg->obs_h == 64 ? 8 : 6) : cfg->obs_h;
    e->model->HWl = e->model->GW * e->model->GH;
    return LZ_OK;
}

static int refresh_materialize_raw(lz_engine *e);

extern "C" int lz_model_set_tensor(lz_engine *e, const char *name, const float *h_data, const int64_t *shape, int ndim)
{
    LZ_REQUIRE(e != nullptr && e->model != nullptr, "no model: call lz_model_create first");
    LZ_REQUIRE(name && h_data && (shape || ndim == 0) && ndim >= 0 && ndim <= 4, "bad tensor argument");
    if (e->model->raw_stale) { if (int rc = refresh_materialize_raw(e)) return rc; }
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    t.data.assign(h_data, h_data + n);
    e->model->raw[name] = std::move(t);
    e->model->finalized = false;
    return LZ_OK;
}

// The same from a DEVICE tensor (a weight refresh whose state_dict arrived by an RCCL broadcast: shard.broadcast_state_dict(...,
// on_device=True)): the copy into the host-side staging map happens here, on the engine's stream, instead of as one
// flat.cpu().numpy() + per-tensor slicing in the caller.  The re-layout of lz_model_finalize itself is host code.
// The caller has made sure the producing stream is done with d_data.
extern "C" int lz_model_set_tensor_device(lz_engine *e, const char *name, const float *d_data, const int64_t *shape, int ndim)
{
    LZ_REQUIRE(e != nullptr && e->model != nullptr, "no model: call lz_model_create first");
    LZ_REQUIRE(name && d_data && (shape || ndim == 0) && ndim >= 0 && ndim <= 4, "bad tensor argument");
    LZ_HIP_CHECK(hipSetDevice(e->device));
    if (e->model->raw_stale) { if (int rc = refresh_materialize_raw(e)) return rc; }
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    t.data.resize(n);
    LZ_HIP_CHECK(hipMemcpyAsync(t.data.data(), d_data, n * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    LZ_HIP_CHECK(hipStreamSynchronize(e->stream));   // (pageable destination; the vector is moved into the map below)
    e->model->raw[name] = std::move(t);
    e->model->finalized = false;
    return LZ_OK;
}

static void finalize_conv_layouts(lz_model *m, Builder &b);
static void refresh_program_free(lz_model *m);

extern "C" int lz_model_finalize(lz_engine *e)
{
    LZ_REQUIRE(e != nullptr && e->model != nullptr, "no model: call lz_model_create first");
    LZ_HIP_CHECK(hipSetDevice(e->device));
    lz_model *m = e->model;
    // a weight refresh on a live engine: nothing of this engine may still be reading the buffers that are about to be
    // overwritten in place (Builder::upload)
    LZ_HIP_CHECK(hipStreamSynchronize(e->stream));
    m->alloc_cursor = 0;
    m->realloc_happened = false;
    struct GenBump {  // on every exit path: re-allocated weights invalidate the captured graphs of this engine's roots
        lz_engine *e; lz_model *m;
        ~GenBump()
        {
            if (m->alloc_cursor < m->allocs.size()) {  // fewer buffers than last time
                for (size_t i = m->alloc_cursor; i < m->allocs.size(); ++i) (void)hipFree(m->allocs[i]);
                m->allocs.resize(m->alloc_cursor); m->alloc_bytes.resize(m->alloc_cursor);
                m->realloc_happened = true;
            }
            if (m->realloc_happened) e->weights_gen++;
        }
    } gen_bump{e, m};
    if (m->cfg.model_type >= 2) return lz_mlp_finalize(e);
    Builder b{m, ""};
    finalize_conv_layouts(m, b);
    if (!b.err.empty()) {
        lz_set_error("lz_model_finalize: %s", b.err.c_str());
        return LZ_ERR_STATE;
    }
    LZ_HIP_CHECK(hipDeviceSynchronize());  // weight uploads went through the null stream; the engine stream does not order against it
    m->finalized = true;
    m->raw_stale = false;
    if (m->refresh.tried && m->refresh.n_allocs != m->allocs.size()) refresh_program_free(m);   // (other shapes: recorded for other buffers)
    if (m->realloc_happened) refresh_program_free(m);
    return LZ_OK;
}

// every device layout of a convolutional model's weights, in a fixed upload order (Builder::upload).  Runs in two modes: on the
// state_dict's values (lz_model_finalize) and on element codes (Builder::rec: the recording pass of the device-side refresh, lz_model.h)
static void finalize_conv_layouts(lz_model *m, Builder &b)
{
    const lz_model_cfg &c = m->cfg;
    const int C = c.num_channels, C2 = C / 2, A = c.action_space_size, HC = c.head_channels, HID = c.head_hidden,
              H = c.lstm_hidden_size, HW = m->HWl, SUP = c.support_size, NRB = c.num_res_blocks > 0 ? c.num_res_blocks : 1;
    const int RSUP = c.reward_support_size > 0 ? c.reward_support_size : SUP;
    const int AE = c.action_encoding == 1 ? 1 : A;   // action planes of the dynamics convolution's input (efficientzero_model.py:105-108)
    const bool wchain = C == 64 && ((m->GW == 6 && m->GH == 6) || (m->GW == 8 && m->GH == 8));  // these chains run on Winograd-transformed weights (k_chain_w)
    // parity mode: the chains' weights as three exact bf16 planes (k_chain_s3 on 6x6, k_chain_s3g on the other grids with an instance)
    const bool s3chain = C == 64 && c.precision == 0 && ((m->GW == 6 && m->GH == 6) || lz_chain_s3g_supported(m->GW, m->GH, false, false));
    // ---- representation (common.py:266-365, :706-787)
    {
        if (!c.downsample) {
            // board games: conv3x3(obs_c -> C) + BN + ReLU (common.py:735-741)
            const HostTensor *w = b.get("representation_network.conv.weight", {C, c.obs_c, 3, 3});
            std::vector<float> sc, sh;
            b.bn("representation_network.norm", C, sc, sh);
            if (w) {
                std::vector<float> p((size_t)9 * c.obs_c * C);  // [tap][ci][co]
                for (int co = 0; co < C; ++co)
                    for (int ci = 0; ci < c.obs_c; ++ci)
                        for (int t = 0; t < 9; ++t) p[((size_t)t * c.obs_c + ci) * C + co] = w->data[((size_t)co * c.obs_c + ci) * 9 + t];
                m->rin.w = b.upload(p);
                m->rin.scale = b.upload(sc);
                m->rin.shift = b.upload(sh);
            }
        } else {
        const std::string d = "representation_network.downsample_net.";
        const HostTensor *w = b.get(d + "conv1.weight", {C2, c.obs_c, 3, 3});
        std::vector<float> sc, sh;
        b.bn(d + "norm1", C2, sc, sh);
        if (w) {
            std::vector<float> p((size_t)9 * c.obs_c * C2);  // [tap][ci][co]
            for (int co = 0; co < C2; ++co)
                for (int ci = 0; ci < c.obs_c; ++ci)
                    for (int t = 0; t < 9; ++t) p[((size_t)t * c.obs_c + ci) * C2 + co] = w->data[((size_t)co * c.obs_c + ci) * 9 + t];
            m->first_w = b.upload(p);
            m->first_s = b.upload(sc);
            m->first_t = b.upload(sh);
        }
        m->r1a = b.resconv(d + "resblocks1.0", 1, C2, C2, true);
        m->r1b = b.resconv(d + "resblocks1.0", 2, C2, C2, true);
        m->dn1 = b.resconv(d + "downsample_block", 1, C, C2);
        m->dn2 = b.resconv(d + "downsample_block", 2, C, C, true);
        m->dn3 = b.conv(d + "downsample_block.conv3.0.weight", "", C, C2, C2);
        m->r2a = b.resconv(d + "resblocks2.0", 1, C, C, true);
        m->r2b = b.resconv(d + "resblocks2.0", 2, C, C, true);
        m->r3a = b.resconv(d + "resblocks3.0", 1, C, C, true);
        m->r3b = b.resconv(d + "resblocks3.0", 2, C, C, true);
        if (c.precision == 0 && C == 64) {   // parity mode: the tower's convolutions as split-bf16 products (k_conv_s3: fp32 accuracy on the bf16 matrix pipe)
            auto rc = [&](const std::string &blk, int idx) { return d + blk + ".conv" + std::to_string(idx) + ".0.weight"; };
            b.split3_tower(rc("resblocks1.0", 1), C2, C2, m->r1a); b.split3_tower(rc("resblocks1.0", 2), C2, C2, m->r1b);
            b.split3_tower(rc("downsample_block", 1), C, C2, m->dn1); b.split3_tower(rc("downsample_block", 2), C, C, m->dn2);
            b.split3_tower(d + "downsample_block.conv3.0.weight", C, C2, m->dn3);
            b.split3_tower(rc("resblocks2.0", 1), C, C, m->r2a); b.split3_tower(rc("resblocks2.0", 2), C, C, m->r2b);
            b.split3_tower(rc("resblocks3.0", 1), C, C, m->r3a); b.split3_tower(rc("resblocks3.0", 2), C, C, m->r3b);
        }
        if (c.precision == 1) {   // fast mode: the tower's convolutions on bf16 MFMA (k_conv_bf)
            auto rc = [&](const std::string &blk, int idx) { return d + blk + ".conv" + std::to_string(idx) + ".0.weight"; };
            m->r1a.wt = b.bf16_tower(rc("resblocks1.0", 1), C2, C2); m->r1b.wt = b.bf16_tower(rc("resblocks1.0", 2), C2, C2);
            m->dn1.wt = b.bf16_tower(rc("downsample_block", 1), C, C2); m->dn2.wt = b.bf16_tower(rc("downsample_block", 2), C, C);
            m->dn3.wt = b.bf16_tower(d + "downsample_block.conv3.0.weight", C, C2);
            m->r2a.wt = b.bf16_tower(rc("resblocks2.0", 1), C, C); m->r2b.wt = b.bf16_tower(rc("resblocks2.0", 2), C, C);
            m->r3a.wt = b.bf16_tower(rc("resblocks3.0", 1), C, C); m->r3b.wt = b.bf16_tower(rc("resblocks3.0", 2), C, C);
        }
        }
        m->rep_res.clear();
        for (int i = 0; i < NRB; ++i) {
            const std::string p = "representation_network.resblocks." + std::to_string(i);
            m->rep_res.push_back(b.resconv(p, 1, C, C, false, wchain, s3chain));
            m->rep_res.push_back(b.resconv(p, 2, C, C, false, wchain, s3chain));
        }
    }
    // ---- dynamics (efficientzero_model.py:427-569)
    {
        const std::string d = "dynamics_network.";
        m->dyn = b.conv(d + "conv.weight", d + "norm_common", C, C + AE, C);
        if (wchain) m->dyn.uc = b.wino_chain(d + "conv.weight", C, C + AE, C);
        if (wchain && c.precision == 1) m->dyn.wb = b.bf16_chain(d + "conv.weight", C, C + AE, C);
        if (s3chain) b.split3_chain(d + "conv.weight", C, C + AE, C, m->dyn);
        // one-hot action planes: plane a is all ones inside the 6x6 latent, so its contribution to output
        // (pixel p, channel co) is the sum of W[co][C+a][tap] over the taps that stay inside the image.  not_one_hot: ONE plane holding
        // action / action_space_size (fp32, like the reference's expand(...) / A): entry a = the in-bounds taps of W[co][C] times that
        const HostTensor *w = b.get(d + "conv.weight", {C, C + AE, 3, 3});
        if (w) {
            const int SW = m->GW, SH = m->GH;
            std::vector<float> tab((size_t)A * HW * C);
            if (b.rec) {   // derived tensor, already in the table's own order (k_refresh_act evaluates the loop below)
                const int64_t o = b.rec->derived((int64_t)tab.size());
                b.rec->act.push_back(RefreshRec::Act{b.src0(w), A, AE, C, SW, SH, (int32_t)o});
                for (size_t i = 0; i < tab.size(); ++i) tab[i] = RefreshRec::code(o + (int64_t)i);
            } else
            for (int a = 0; a < A; ++a)
                for (int y = 0; y < SH; ++y)
                    for (int x = 0; x < SW; ++x)
                        for (int co = 0; co < C; ++co) {
                            float acc = 0.0f;
                            const float plane = AE == 1 ? (float)a / (float)A : 1.0f;
                            for (int t = 0; t < 9; ++t) {
                                const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
                                if (iy >= 0 && iy < SH && ix >= 0 && ix < SW) acc += w->data[((size_t)co * (C + AE) + C + (AE == 1 ? 0 : a)) * 9 + t] * plane;
                            }
                            tab[((size_t)a * HW + y * SW + x) * C + co] = acc;
                        }
            m->act_table = b.upload(tab);
        }
        m->dyn_res.clear();
        for (int i = 0; i < NRB; ++i) {
            m->dyn_res.push_back(b.resconv(d + "resblocks." + std::to_string(i), 1, C, C, false, wchain, s3chain));
            m->dyn_res.push_back(b.resconv(d + "resblocks." + std::to_string(i), 2, C, C, false, wchain, s3chain));
        }
        m->rew_c = b.conv1x1(d + "conv1x1_reward", d + "norm_reward", HC, C);
        if (c.model_type == 1) {
            // MuZero DynamicsNetwork (muzero_model.py:505-538): reward = MLP(flatten(relu(bn(conv1x1(next latent)))))
            m->fc_reward = b.mlp(d + "fc_reward_head", HC * HW, HID, RSUP, true, HC, HW);
        } else {
        // LSTM: rows re-ordered to 4*unit + gate; the x columns permuted from the reference's
        // (channel, pixel) flatten order to (pixel, channel)
        const int KX = HC * HW, K = KX + H;
        const HostTensor *wih = b.get(d + "lstm.weight_ih_l0", {4 * H, KX}), *whh = b.get(d + "lstm.weight_hh_l0", {4 * H, H}),
                         *bih = b.get(d + "lstm.bias_ih_l0", {4 * H}), *bhh = b.get(d + "lstm.bias_hh_l0", {4 * H});
        if (wih && whh && bih && bhh) {
            std::vector<float> wc((size_t)4 * H * K), bc((size_t)4 * H);
            const int64_t o_bias = b.rec ? b.rec->derived(4 * (int64_t)H) : 0;   // derived tensor bias_ih + bias_hh in the reference's row order
            if (b.rec) b.rec->add.push_back(RefreshRec::Add{b.src0(bih), b.src0(bhh), 4 * H, (int32_t)o_bias});
            for (int g = 0; g < 4; ++g)
                for (int u = 0; u < H; ++u) {
                    const int src = g * H + u, dst = 4 * u + g;
                    for (int p = 0; p < HW; ++p)
                        for (int ch = 0; ch < HC; ++ch) wc[(size_t)dst * K + p * HC + ch] = wih->data[(size_t)src * KX + ch * HW + p];
                    for (int k = 0; k < H; ++k) wc[(size_t)dst * K + KX + k] = whh->data[(size_t)src * H + k];
                    bc[dst] = b.rec ? RefreshRec::code(o_bias + src) : bih->data[src] + bhh->data[src];
                }
            m->lstm_w = b.upload(wc);
            m->lstm_b = b.upload(bc);
            if (K % 16 == 0 && H % 16 == 0) {
                std::vector<float> wf(wc.size());
                lz_lstm_pack_fragments(wc.data(), H, K, wf.data());
                m->lstm_wf = b.upload(wf);
            }
            m->lstm_wb = nullptr;
            if (c.precision == 1 && (KX == 576 || KX == 1024) && H == 512) {
                // k_lstm_b: wave = gate g of unit tile t, lane (n = l & 15, kq = l >> 4) holds W[4 (16 t + n) + g][32 s + 8 kq + j], j = 0..7
                const int NS = K / 32;
                std::vector<uint16_t> wb((size_t)4 * H * K);
                for (int t = 0; t < H / 16; ++t)
                    for (int g = 0; g < 4; ++g)
                        for (int st = 0; st < NS; ++st)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int j = 0; j < 8; ++j)
                                    wb[((((size_t)(t * 4 + g) * NS + st) * 64) + lane) * 8 + j] =
                                        Builder::bf16_rne(wc[(size_t)(4 * (16 * t + (lane & 15)) + g) * K + 32 * st + 8 * (lane >> 4) + j]);
                m->lstm_wb = b.upload_u16(wb);
            }
        }
        std::vector<float> sc, sh;
        b.bn(d + "norm_value_prefix", H, sc, sh);
        m->vp_s = b.upload(sc);
        m->vp_t = b.upload(sh);
        if (c.num_of_sampled_actions > 0) m->wh_reward = b.wide_head(d + "fc_reward_head", H, HID, SUP, false, HC, HW);   // conv Sampled EfficientZero
        else m->fc_reward = b.mlp(d + "fc_reward_head", H, HID, SUP, false, HC, HW);
        }
    }
    // ---- prediction (common.py:1081-1216)
    {
        const std::string d = "prediction_network.";
        m->pred_res.clear();
        for (int i = 0; i < NRB; ++i) {
            m->pred_res.push_back(b.resconv(d + "resblocks." + std::to_string(i), 1, C, C, false, wchain, s3chain));
            m->pred_res.push_back(b.resconv(d + "resblocks." + std::to_string(i), 2, C, C, false, wchain, s3chain));
        }
        m->val_c = b.conv1x1(d + "conv1x1_value", d + "norm_value", HC, C);
        m->pol_c = b.conv1x1(d + "conv1x1_policy", d + "norm_policy", HC, C);
        m->wide_heads = c.model_type == 0 && c.num_of_sampled_actions > 0;
        if (m->wide_heads) {   // conv Sampled EfficientZero: hidden width up to 256, ReLU | GELU -> dense-layer kernels
            m->wh_value = b.wide_head(d + "fc_value", HC * HW, HID, SUP, true, HC, HW);
            m->wh_policy = b.wide_head(d + "fc_policy", HC * HW, HID, A, true, HC, HW);
        } else {
        m->fc_value = b.mlp(d + "fc_value", HC * HW, HID, SUP, true, HC, HW);
        m->fc_policy = b.mlp(d + "fc_policy", HC * HW, HID, A, true, HC, HW);
        }
    }
    // ---- split heads (see enqueue_search): first layers of the three head MLPs as per-unit-tile MFMA fragments for the LSTM launch
    m->sh_w1c = m->sh_w1r = nullptr;
    // (8x8 latent, round 6: parity mode only -- k_chain_s3g's split-head instance; the fast-mode and fp32 8x8 chains keep the head launch)
    if (c.model_type == 0 && wchain && (m->GW == 6 || (m->GW == 8 && c.precision == 0)) && HC == 16 && H == 512 && m->fc_value.h_w1.size() == (size_t)32 * HC * HW &&
        m->fc_policy.h_w1.size() == (size_t)32 * HC * HW && m->fc_reward.h_w1.size() == (size_t)32 * H) {
        const int NU = H / 16, K1 = HC * HW, KS = 2 * K1 / NU;   // 32 unit tiles; 576 | 1024; 36 | 64 combined input columns per tile
        if (KS == 36 || KS == 64) {
            // per unit tile u and wave (= 16-column tile nt of the 64 | 32 hidden columns): lane (n = lane & 15, kq = lane >> 4) holds the
            // B operands of its 9 (16 on the 8x8 latent) | 4 k-steps contiguously (12 (16) | 4 floats: three (four) | one float4 load)
            const int SB = KS == 36 ? 12 : 16;
            std::vector<float> wc((size_t)NU * 4 * 64 * SB, 0.0f), wr((size_t)NU * 2 * 64 * 4);
            for (int u = 0; u < NU; ++u) {
                for (int nt = 0; nt < 4; ++nt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int ks = 0; ks < KS / 4; ++ks) {
                            const int kk = KS * u + 4 * ks + (lane >> 4), pix = kk / (2 * HC), c32 = kk % (2 * HC), col = 16 * nt + (lane & 15);
                            float w = 0.0f;
                            if (col < 32 && c32 < HC) w = m->fc_value.h_w1[(size_t)col * K1 + pix * HC + c32];
                            if (col >= 32 && c32 >= HC) w = m->fc_policy.h_w1[(size_t)(col - 32) * K1 + pix * HC + (c32 - HC)];
                            wc[(((size_t)u * 4 + nt) * 64 + lane) * SB + ks] = w;
                        }
                for (int nt = 0; nt < 2; ++nt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int ks = 0; ks < 4; ++ks)
                            wr[(((size_t)u * 2 + nt) * 64 + lane) * 4 + ks] = m->fc_reward.h_w1[(size_t)(16 * nt + (lane & 15)) * H + 16 * u + 4 * ks + (lane >> 4)];
            }
            m->sh_w1c = b.upload(wc);
            m->sh_w1r = b.upload(wr);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Device-side weight refresh (RefreshRec / RefreshProgram in lz_model.h): the derived-tensor kernels and the gather.
// The arithmetic below is the host packers' (Builder::bn, Builder::wino_u, the action table and the LSTM bias of
// finalize_conv_layouts), operation by operation, with FMA contraction off: tests/test_weight_refresh_gpu.py holds the refreshed
// buffers bit-equal to a host finalize of the same state_dict.
#pragma clang fp contract(off)
__global__ void k_refresh_bn(float *src, const RefreshRec::Bn *ops)
{
    const RefreshRec::Bn o = ops[blockIdx.x];
    for (int i = threadIdx.x; i < o.n; i += blockDim.x) {
        const float inv = 1.0f / sqrtf(src[o.var + i] + o.eps);
        const float scale = src[o.w + i] * inv;
        src[o.out + i] = scale;
        src[o.out + o.n + i] = src[o.b + i] - src[o.mu + i] * scale;
    }
}
__global__ void k_refresh_add(float *src, const RefreshRec::Add *ops)
{
    const RefreshRec::Add o = ops[blockIdx.x];
    for (int i = threadIdx.x; i < o.n; i += blockDim.x) src[o.out + i] = src[o.a + i] + src[o.b + i];
}
// U = G g G^T per (cout, cin) filter in binary64, rounded once (Builder::wino_u); one thread per filter
__global__ void k_refresh_wino(float *src, const RefreshRec::Wino *ops)
{
    const RefreshRec::Wino o = ops[blockIdx.y];
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= o.cout * o.cin) return;
    const int co = f / o.cin, ci = f - co * o.cin;
    const float *g = src + o.w + ((size_t)co * o.cin_total + ci) * 9;
    const double Gm[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    double t[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) t[i][k] = Gm[i][0] * g[0 * 3 + k] + Gm[i][1] * g[1 * 3 + k] + Gm[i][2] * g[2 * 3 + k];
    float *out = src + o.out + (size_t)f * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) out[4 * i + j] = (float)(t[i][0] * Gm[j][0] + t[i][1] * Gm[j][1] + t[i][2] * Gm[j][2]);
}
// the one-hot (or single-plane) action table: sum of the in-bounds taps of the action's input plane, per (action, pixel, channel)
__global__ void k_refresh_act(float *src, const RefreshRec::Act *ops)
{
    const RefreshRec::Act o = ops[blockIdx.y];
    const int HW = o.SW * o.SH, n = o.A * HW * o.C;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int co = i % o.C, pix = (i / o.C) % HW, a = i / (o.C * HW), y = pix / o.SW, x = pix - y * o.SW;
    float acc = 0.0f;
    const float plane = o.AE == 1 ? (float)a / (float)o.A : 1.0f;
    for (int t = 0; t < 9; ++t) {
        const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
        if (iy >= 0 && iy < o.SH && ix >= 0 && ix < o.SW) acc += src[o.w + ((size_t)co * (o.C + o.AE) + o.C + (o.AE == 1 ? 0 : a)) * 9 + t] * plane;
    }
    src[o.out + i] = acc;
}
// a gathered fp32 fragment buffer -> its three bf16 planes (Builder::split3_tower: hi = rne(w), mid = rne(w - hi), lo = rne(w - hi - mid)),
// blocks of 64 lanes x 8: [block][plane][512]
__global__ void k_refresh_split3(const RefreshRec::Split3 *ops)
{
    const RefreshRec::Split3 o = ops[blockIdx.y];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= o.n) return;
    const float w = o.src[i];
    const __bf16 h = (__bf16)w;
    const float r1 = w - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    const __bf16 l = (__bf16)r2;
    __bf16 *dst = reinterpret_cast<__bf16 *>(o.dst);
    const int64_t blk = i >> 9, e = i & 511;
    dst[(blk * 3 + 0) * 512 + e] = h;
    dst[(blk * 3 + 1) * 512 + e] = m;
    dst[(blk * 3 + 2) * 512 + e] = l;
}
#pragma clang fp contract(fast)
// every weight buffer from the source space: slot s covers idx[start .. start + count)
__global__ void k_refresh_gather(const float *__restrict__ src, const int32_t *__restrict__ idx, const RefreshRec::Slot *__restrict__ slots, int n_slots, int64_t total)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int lo = 0, hi = n_slots - 1;   // the slot that holds element i
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (slots[mid].start <= i) lo = mid; else hi = mid - 1;
    }
    const int32_t k = idx[i];
    slots[lo].dst[i - slots[lo].start] = k < 0 ? 0.0f : src[k];
}

static void refresh_program_free(lz_model *m)
{
    RefreshProgram &p = m->refresh;
    for (void *q : {(void *)p.d_src, (void *)p.d_idx, p.d_slots, p.d_bn, p.d_wino, p.d_act, p.d_add, p.d_split3}) if (q) (void)hipFree(q);
    if (p.h_pin) (void)hipHostFree(p.h_pin);
    p = RefreshProgram{};
}

static int refresh_run(lz_engine *e, const float *flat, int64_t n_floats, int on_device);
extern "C" int lz_model_weights_digest(lz_engine *e, uint64_t *out);
// the recording pass: finalize_conv_layouts once more, on element codes (see RefreshRec)
static int refresh_program_build(lz_engine *e)
{
    lz_model *m = e->model;
    RefreshProgram &p = m->refresh;
    refresh_program_free(m);
    p.tried = true;
    p.n_allocs = m->allocs.size();
    if (m->cfg.model_type >= 2) { p.why = "vector-observation (MLP) models re-lay their weights out on the host"; return LZ_OK; }
    if (m->cfg.precision == 1) { p.why = "fast mode (bf16 fragments) re-lays its weights out on the host"; return LZ_OK; }
    if (!m->finalized || m->raw_stale) { p.why = "no finalized host copy of the weights to record from"; return LZ_OK; }
    RefreshRec rec;
    std::map<std::string, HostTensor> shadow;
    for (const auto &kv : m->raw) {   // std::map order == sorted names == the flat layout
        HostTensor t;
        t.shape = kv.second.shape;
        const int64_t n = (int64_t)kv.second.data.size();
        rec.names.push_back(kv.first); rec.offsets.push_back(rec.raw_floats); rec.sizes.push_back(n);
        t.data.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i) t.data[(size_t)i] = RefreshRec::code(1 + rec.raw_floats + i);
        rec.raw_floats += n;
        shadow[kv.first] = std::move(t);
    }
    m->raw.swap(shadow);
    const size_t cursor = m->alloc_cursor;
    const bool realloc_flag = m->realloc_happened;
    m->alloc_cursor = 0;
    Builder b{m, ""};
    b.rec = &rec;
    finalize_conv_layouts(m, b);
    m->raw.swap(shadow);
    if (m->alloc_cursor != m->allocs.size()) rec.fail("the recording pass produced another number of weight buffers");
    m->alloc_cursor = cursor;
    m->realloc_happened = realloc_flag;
    if (!b.err.empty()) rec.fail(b.err);
    const int64_t src_floats = 1 + rec.raw_floats + rec.derived_floats;
    if (src_floats + 1 >= (int64_t)1 << 24) rec.fail("more than 2^24 source elements: codes are no longer exact in binary32");
    if (!rec.ok) { p.why = rec.why; return LZ_OK; }
    p.names = rec.names; p.offsets = rec.offsets; p.sizes = rec.sizes;
    p.raw_floats = rec.raw_floats; p.src_floats = src_floats; p.out_floats = (int64_t)rec.idx.size();
    auto up = [&](void **d, const void *h, size_t bytes) -> bool {
        if (!bytes) return true;
        if (lz_dev_malloc(d, bytes) != hipSuccess) return false;
        return hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice) == hipSuccess;
    };
    bool ok = lz_dev_malloc((void **)&p.d_src, (size_t)src_floats * 4) == hipSuccess;
    const float one = 1.0f;
    ok = ok && hipMemcpy(p.d_src, &one, 4, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && up((void **)&p.d_idx, rec.idx.data(), rec.idx.size() * 4);
    ok = ok && up(&p.d_slots, rec.slots.data(), rec.slots.size() * sizeof(RefreshRec::Slot));
    ok = ok && up(&p.d_bn, rec.bn.data(), rec.bn.size() * sizeof(RefreshRec::Bn));
    ok = ok && up(&p.d_wino, rec.wino.data(), rec.wino.size() * sizeof(RefreshRec::Wino));
    ok = ok && up(&p.d_act, rec.act.data(), rec.act.size() * sizeof(RefreshRec::Act));
    ok = ok && up(&p.d_add, rec.add.data(), rec.add.size() * sizeof(RefreshRec::Add));
    ok = ok && up(&p.d_split3, rec.split3.data(), rec.split3.size() * sizeof(RefreshRec::Split3));
    if (!ok) { refresh_program_free(m); p.tried = true; p.why = "out of device memory for the refresh program"; return LZ_OK; }
    p.n_slots = (int)rec.slots.size(); p.n_bn = (int)rec.bn.size(); p.n_wino = (int)rec.wino.size(); p.n_act = (int)rec.act.size(); p.n_add = (int)rec.add.size();
    for (const auto &o : rec.wino) p.wino_items = std::max<int64_t>(p.wino_items, (int64_t)o.cout * o.cin);
    for (const auto &o : rec.act) p.act_items = std::max<int64_t>(p.act_items, (int64_t)o.A * o.SW * o.SH * o.C);
    p.n_split3 = (int)rec.split3.size();
    for (const auto &o : rec.split3) p.split3_items = std::max<int64_t>(p.split3_items, o.n);
    p.n_allocs = m->allocs.size();
    p.usable = true;
    // Self-check: the program was RECORDED by running the host packers on element codes -- a packer that did arithmetic on two codes
    // and still produced an integer in range would have recorded a wrong gather map silently.  Run the program once on the weights that are
    // loaded right now (the host copy the packers just read) and require every device weight buffer to come out byte for byte as the host
    // finalize left it; otherwise the program is unusable and the host finalize restores the buffers.
    {
        uint64_t d0 = 0, d1 = 0;
        if (int rc = lz_model_weights_digest(e, &d0)) return rc;
        std::vector<float> flat((size_t)p.raw_floats);
        size_t i = 0;
        for (const auto &kv : m->raw) {
            memcpy(flat.data() + p.offsets[i], kv.second.data.data(), (size_t)p.sizes[i] * 4);
            ++i;
        }
        int rc = refresh_run(e, flat.data(), p.raw_floats, 0);
        m->raw_stale = false;              // (the flat buffer WAS the host copy)
        if (!rc) rc = lz_model_weights_digest(e, &d1);
        if (rc || d0 != d1) {
            refresh_program_free(m);
            m->refresh.tried = true;
            m->refresh.why = rc ? "the refresh program's self-check could not run" : "the recorded refresh program does not reproduce the host re-layout of the loaded weights (self-check)";
            (void)lz_model_finalize(e);    // the host re-layout writes every buffer again
            m->refresh.tried = true;
            return LZ_OK;
        }
    }
    return LZ_OK;
}

static int refresh_ready(lz_engine *e)
{
    LZ_REQUIRE(e != nullptr && e->model != nullptr, "no model: call lz_model_create first");
    LZ_REQUIRE(e->model->finalized, "the model has not been loaded yet: the first load goes through lz_model_set_tensor + lz_model_finalize");
    LZ_HIP_CHECK(hipSetDevice(e->device));
    if (!e->model->refresh.tried) { if (int rc = refresh_program_build(e)) return rc; }
    if (!e->model->refresh.usable) {
        lz_set_error("no device-side refresh for this model: %s", e->model->refresh.why.c_str());
        return LZ_ERR_STATE;
    }
    return LZ_OK;
}

extern "C" int lz_model_flat_layout(lz_engine *e, int64_t *out_tensors, int64_t *out_floats)
{
    if (int rc = refresh_ready(e)) return rc;
    if (out_tensors) *out_tensors = (int64_t)e->model->refresh.names.size();
    if (out_floats) *out_floats = e->model->refresh.raw_floats;
    return LZ_OK;
}

extern "C" int lz_model_flat_entry(lz_engine *e, int64_t i, char *name_buf, int64_t name_buf_len, int64_t *out_offset, int64_t *out_size)
{
    if (int rc = refresh_ready(e)) return rc;
    const RefreshProgram &p = e->model->refresh;
    LZ_REQUIRE(i >= 0 && i < (int64_t)p.names.size() && name_buf && name_buf_len > 0, "entry index out of range");
    snprintf(name_buf, (size_t)name_buf_len, "%s", p.names[(size_t)i].c_str());
    LZ_REQUIRE((int64_t)p.names[(size_t)i].size() < name_buf_len, "name buffer too short");
    if (out_offset) *out_offset = p.offsets[(size_t)i];
    if (out_size) *out_size = p.sizes[(size_t)i];
    return LZ_OK;
}

// the pinned host staging buffer of the flat state_dict (raw_floats floats), free to be written: the upload of the previous refresh out
// of it has completed when this returns
extern "C" int lz_model_flat_host_buffer(lz_engine *e, float **out)
{
    if (int rc = refresh_ready(e)) return rc;
    RefreshProgram &p = e->model->refresh;
    LZ_REQUIRE(out != nullptr, "NULL argument");
    if (!p.h_pin) LZ_HIP_CHECK(hipHostMalloc(&p.h_pin, (size_t)p.raw_floats * 4, hipHostMallocDefault));
    else LZ_HIP_CHECK(hipStreamSynchronize(e->stream));
    *out = (float *)p.h_pin;
    return LZ_OK;
}

// flat: the model's state_dict tensors (without num_batches_tracked), fp32, concatenated in name order (lz_model_flat_entry) -- a device
// pointer (on_device != 0: e.g. the buffer an RCCL broadcast filled; the caller has made sure its producer is done) or a host pointer.
// Everything is enqueued on the engine's stream: searches launched before see the old weights, searches launched after the new ones.
extern "C" int lz_model_refresh_flat(lz_engine *e, const float *flat, int64_t n_floats, int on_device)
{
    if (int rc = refresh_ready(e)) return rc;
    return refresh_run(e, flat, n_floats, on_device);
}
static int refresh_run(lz_engine *e, const float *flat, int64_t n_floats, int on_device)
{
    lz_model *m = e->model;
    RefreshProgram &p = m->refresh;
    LZ_REQUIRE(flat != nullptr && n_floats == p.raw_floats, "flat state_dict of another size than the model's tensors");
    hipStream_t s = e->stream;
    if (on_device) {
        LZ_HIP_CHECK(hipMemcpyAsync(p.d_src + 1, flat, (size_t)n_floats * 4, hipMemcpyDeviceToDevice, s));
    } else {
        // through pinned staging, so that the copy is one asynchronous DMA (a pageable source is staged piecewise by the runtime).  A
        // caller that filled the staging buffer itself (lz_model_flat_host_buffer) passes its address: no second copy.
        if (flat != (const float *)p.h_pin) {
            float *pin = nullptr;
            if (int rc = lz_model_flat_host_buffer(e, &pin)) return rc;   // (waits for the previous upload out of this buffer)
            memcpy(pin, flat, (size_t)n_floats * 4);
        }
        LZ_HIP_CHECK(hipMemcpyAsync(p.d_src + 1, p.h_pin, (size_t)n_floats * 4, hipMemcpyHostToDevice, s));
    }
    if (p.n_bn) hipLaunchKernelGGL(k_refresh_bn, dim3((unsigned)p.n_bn), dim3(256), 0, s, p.d_src, (const RefreshRec::Bn *)p.d_bn);
    if (p.n_add) hipLaunchKernelGGL(k_refresh_add, dim3((unsigned)p.n_add), dim3(256), 0, s, p.d_src, (const RefreshRec::Add *)p.d_add);
    if (p.n_wino) hipLaunchKernelGGL(k_refresh_wino, dim3((unsigned)((p.wino_items + 255) / 256), (unsigned)p.n_wino), dim3(256), 0, s, p.d_src, (const RefreshRec::Wino *)p.d_wino);
    if (p.n_act) hipLaunchKernelGGL(k_refresh_act, dim3((unsigned)((p.act_items + 255) / 256), (unsigned)p.n_act), dim3(256), 0, s, p.d_src, (const RefreshRec::Act *)p.d_act);
    hipLaunchKernelGGL(k_refresh_gather, dim3((unsigned)((p.out_floats + 255) / 256)), dim3(256), 0, s, p.d_src, p.d_idx, (const RefreshRec::Slot *)p.d_slots, p.n_slots, p.out_floats);
    if (p.n_split3) hipLaunchKernelGGL(k_refresh_split3, dim3((unsigned)((p.split3_items + 255) / 256), (unsigned)p.n_split3), dim3(256), 0, s, (const RefreshRec::Split3 *)p.d_split3);
    LZ_HIP_CHECK(hipGetLastError());
    m->raw_stale = true;
    return LZ_OK;
}

// `raw` (the host copy lz_model_finalize reads) after device-side refreshes: read it back from the source buffer
static int refresh_materialize_raw(lz_engine *e)
{
    lz_model *m = e->model;
    RefreshProgram &p = m->refresh;
    LZ_REQUIRE(p.usable && p.d_src, "the host copy of the weights is stale and there is no source buffer to restore it from");
    std::vector<float> flat((size_t)p.raw_floats);
    LZ_HIP_CHECK(hipMemcpyAsync(flat.data(), p.d_src + 1, flat.size() * 4, hipMemcpyDeviceToHost, e->stream));
    LZ_HIP_CHECK(hipStreamSynchronize(e->stream));
    for (size_t i = 0; i < p.names.size(); ++i) {
        auto it = m->raw.find(p.names[i]);
        if (it == m->raw.end() || (int64_t)it->second.data.size() != p.sizes[i]) continue;
        memcpy(it->second.data.data(), flat.data() + p.offsets[i], (size_t)p.sizes[i] * 4);
    }
    m->raw_stale = false;
    return LZ_OK;
}

// FNV-1a over every device weight buffer in upload order (tests: a device-side refresh must leave exactly the bytes a host finalize
// of the same state_dict leaves)
extern "C" int lz_model_weights_digest(lz_engine *e, uint64_t *out)
{
    LZ_REQUIRE(e != nullptr && e->model != nullptr && out != nullptr, "NULL argument");
    LZ_HIP_CHECK(hipSetDevice(e->device));
    LZ_HIP_CHECK(hipStreamSynchronize(e->stream));
    uint64_t h = 1469598103934665603ull;
    std::vector<unsigned char> buf;
    for (size_t i = 0; i < e->model->allocs.size(); ++i) {
        buf.resize(e->model->alloc_bytes[i]);
        if (buf.empty()) continue;
        LZ_HIP_CHECK(hipMemcpy(buf.data(), e->model->allocs[i], buf.size(), hipMemcpyDeviceToHost));
        for (unsigned char c : buf) { h ^= c; h *= 1099511628211ull; }
    }
    *out = h;
    return LZ_OK;
}

// ------------------------------------------------------------------------------------------------
static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// the pools are sized by the model's shapes: a roots handle that outlives its model (another lz_model_create on the engine)
// gets fresh pools, result blocks and observation staging for the new one
int lz_roots_release_pools_if_stale(lz_roots *r)
{
    if (r->pool_model_uid == r->eng->model_uid) return LZ_OK;
    if (r->pool_slab || r->d_obs || r->d_results) {
        LZ_HIP_CHECK(hipStreamSynchronize(r->eng->stream));
        if (r->graph_exec) { (void)hipGraphExecDestroy(r->graph_exec); r->graph_exec = nullptr; }
        if (r->pool_slab) { (void)hipFree(r->pool_slab); r->pool_slab = nullptr; }
        if (r->hd_logits) { (void)hipFree(r->hd_logits); r->hd_logits = r->hd_expect = nullptr; }
        if (r->d_obs) { if (r->last_obs == r->d_obs) r->last_obs = nullptr; (void)hipFree(r->d_obs); r->d_obs = nullptr; }
        if (r->d_results) { (void)hipFree(r->d_results); r->d_results = nullptr; }
        if (r->h_results) { (void)hipHostFree(r->h_results); r->h_results = nullptr; }
        r->d_obs_bytes = 0;
        r->results_bytes = 0;
        r->inferred = false; r->inference_fresh = false;
    }
    r->pool_model_uid = r->eng->model_uid;
    return LZ_OK;
}

// width of a policy row of the model behind these roots: the MLP families say it themselves (2 D for continuous Sampled EfficientZero); a conv
// model's is its action space -- which for the conv Sampled EfficientZero (discrete actions, K sampled per node) is NOT the roots' A = K
static size_t policy_width_of(const lz_model *m, const lz_tree_dev &t)
{
    if (m->cfg.model_type >= 2) return (size_t)lz_mlp_policy_width(m);
    return t.variant == LZ_TREE_SAMPLED_EFFICIENTZERO ? (size_t)m->cfg.action_space_size : (size_t)t.A;
}

static int ensure_pools(lz_roots *r)
{
    if (int rc = lz_roots_release_pools_if_stale(r)) return rc;
    if (r->pool_slab) return LZ_OK;
    lz_model *m = r->eng->model;
    const lz_model_cfg &c = m->cfg;
    const size_t B = r->t.B, NN = r->t.NN, A = std::max((size_t)r->t.A, policy_width_of(m, r->t)), C = c.num_channels, HW = m->HWl,
                 H = c.model_type == 0 ? c.lstm_hidden_size : 0, HC = c.head_channels, SUP = std::max(c.support_size, c.reward_support_size);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_lat = take(NN * B * HW * C * 4), o_h = take(NN * B * H * 4), o_c = take(NN * B * H * 4),
                 o_vp = take(NN * B * 4), o_val = take(NN * B * 4), o_lg = take(NN * B * A * 4),
                 o_x1 = take(B * HW * C * 4), o_x2 = take(B * HW * C * 4), o_x3 = take(B * HW * C * 4),
                 o_rx = take(B * HW * HC * 4), o_pv = take(B * HW * 2 * HC * 4), o_hbn = take(B * H * 4), o_d0 = take(B * SUP * 4), o_d1 = take(B * SUP * 4),
                 o_tr = take(NN * 5 * B * 4), o_z = take(B * 4),
                 o_tp = take((2 * B + B * A) * 4);   // [to_play B | noise offsets B | noise <= B A]: one upload per prepare
    // split heads: first-layer partial sums of the three head MLPs, one block per (16-row tile, LSTM unit tile)
    const size_t NU = H / 16;
    const size_t o_pc = take(m->sh_w1c ? B * 3 * NU * 32 * 4 : 0);
    const size_t fc_bytes = (m->sh_w1c && (B == 256 || B == 128)) ? lz_fused_ctl_bytes((int)B, (int)NN) : 0;   // one launch per simulation (opt-in)
    const size_t o_fc = take(fc_bytes);
    hipError_t err = lz_dev_malloc((void **)&r->pool_slab, off);
    if (err != hipSuccess) {
        lz_set_error("hipMalloc(%zu bytes) for the latent/LSTM pools failed: %s", off, hipGetErrorString(err));
        return err == hipErrorOutOfMemory ? LZ_ERR_NOMEM : LZ_ERR_HIP;
    }
    char *base = (char *)r->pool_slab;
    r->latent_pool = (float *)(base + o_lat); r->h_pool = (float *)(base + o_h); r->c_pool = (float *)(base + o_c);
    r->sim_vp = (float *)(base + o_vp); r->sim_value = (float *)(base + o_val); r->sim_logits = (float *)(base + o_lg);
    r->t_x1 = (float *)(base + o_x1); r->t_x2 = (float *)(base + o_x2); r->t_x3 = (float *)(base + o_x3);
    r->t_rx = (float *)(base + o_rx); r->t_pv = (float *)(base + o_pv); r->t_hbn = (float *)(base + o_hbn);
    r->dbg_logits[0] = (float *)(base + o_d0); r->dbg_logits[1] = (float *)(base + o_d1);
    r->trace = (int32_t *)(base + o_tr); r->d_to_play = (int32_t *)(base + o_tp); r->d_zero_vp = (float *)(base + o_z);
    r->d_noise_off = r->d_to_play + B; r->d_noise = (float *)(r->d_noise_off + B);
    r->sh_part = m->sh_w1c ? (float *)(base + o_pc) : nullptr;
    r->fuse_ctl = fc_bytes ? (void *)(base + o_fc) : nullptr;
    r->fuse_ctl_bytes = fc_bytes;
    LZ_HIP_CHECK(hipMemsetAsync(r->d_zero_vp, 0, B * 4, r->eng->stream));
    return LZ_OK;
}

static int ensure_ws(lz_model *m, int B)
{
    if (m->ws_B >= B) return LZ_OK;
    for (int i = 0; i < 3; ++i) { if (m->ws[i]) (void)hipFree(m->ws[i]); m->ws[i] = nullptr; }
    const lz_model_cfg &c = m->cfg;
    const size_t n = c.downsample ? (size_t)B * (c.obs_h / 2) * (c.obs_w / 2) * (c.num_channels / 2)  // largest activation
                                  : (size_t)B * c.obs_h * c.obs_w * c.num_channels;
    for (int i = 0; i < 3; ++i) LZ_HIP_CHECK(lz_dev_malloc((void **)&m->ws[i], n * 4));
    m->ws_B = B;
    return LZ_OK;
}

// activation codes (1 ReLU, 2 GELU(tanh)) of the dynamics and the prediction network.  Every model here is ReLU throughout, except the conv
// Sampled EfficientZero: its class passes `activation` (default GELU) to the dynamics network ONLY -- the prediction network keeps its own
// default GELU and the representation network its default ReLU (sampled_efficientzero_model.py:177-218; common.py:718)
static int act_dyn(const lz_model_cfg &c) { return (c.model_type == 0 && c.num_of_sampled_actions > 0 && c.activation == 1) ? 2 : 1; }
static int act_pred(const lz_model_cfg &c) { return (c.model_type == 0 && c.num_of_sampled_actions > 0) ? 2 : 1; }

static lz_conv_args conv_args(const ConvW &w, const float *in, float *out, int B, int Hin, int Hout, const float *residual, int relu, int act_bf16)
{
    lz_conv_args a{};
    a.act_bf16 = act_bf16;
    a.in = in; a.w = w.w; a.wf = w.wf; a.uf = w.uf; a.wb = w.wt; a.w3 = w.w3; a.scale = w.scale; a.shift = w.shift; a.residual = residual; a.out = out;
    a.B = B; a.Hin = Hin; a.Win = Hin; a.Hout = Hout; a.Wout = Hout; a.Cout = w.cout; a.relu = relu;
    return a;
}
static void conv(const ConvW &w, const float *in, float *out, int B, int Hin, int Hout, int stride, const float *residual,
                 int relu, hipStream_t s, int act_bf16 = 0)
{
    lz_launch_conv3x3(conv_args(w, in, out, B, Hin, Hout, residual, relu, act_bf16), w.cin, stride, s);
}

__global__ void k_zero_words(unsigned *__restrict__ p, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}

__global__ void k_zero2(float4 *__restrict__ a, float4 *__restrict__ b, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) { a[i] = make_float4(0.f, 0.f, 0.f, 0.f); b[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
}

static lz_chain_layer chlayer(const ConvW &w, int in, int out, int res, int relu, int act, float *gout)
{
    lz_chain_layer l{};
    l.wf = w.wf; l.uc = w.uc; l.wb = w.wb; l.w3 = w.w3c; l.scale = w.scale; l.shift = w.shift; l.in = in; l.out = out; l.res = res; l.relu = relu; l.act = act; l.gout = gout;
    return l;
}

// appends the 2 k convolutions of k residual blocks (ding ResBlock 'basic': conv-bn-relu, conv-bn, + input, relu) to a chain.  x = LDS
// buffer of the input, keep = a buffer that must survive the blocks (-1: none), gout_last = optional HBM copy of the last block's
// output.  Returns the buffer that holds the output.
// `relu` = the activation code of the blocks' network: 1 ReLU, 2 GELU(tanh) (conv Sampled EfficientZero; needs ca.gelu)
static int chain_blocks(lz_chain_args &ca, const std::vector<ConvW> &blocks, int x, int keep, float *gout_last, int relu = 1)
{
    const int k = (int)blocks.size() / 2;
    for (int i = 0; i < k; ++i) {
        int f[2], n = 0;
        for (int b = 0; b < 4 && n < 2; ++b) if (b != x && b != keep) f[n++] = b;
        ca.layer[ca.nlayers++] = chlayer(blocks[2 * i], x, f[0], -1, relu, 0, nullptr);
        ca.layer[ca.nlayers++] = chlayer(blocks[2 * i + 1], f[0], f[1], x, relu, 0, i == k - 1 ? gout_last : nullptr);
        x = f[1];
    }
    return x;
}

static lz_c1_job c1job(const C1W &w, const float *in, float *out, int stride, int off)
{
    lz_c1_job j{};
    j.in = in; j.w = w.w; j.bias = w.b; j.scale = w.s; j.shift = w.t; j.out = out; j.out_stride = stride; j.out_off = off;
    return j;
}

static lz_head_desc headdesc(const MlpW &w, const float *in, int env_stride, int pix_stride, int categorical, float support_min,
                             float *out_logits, float *out_scalar)
{
    lz_head_desc h{};
    h.in = in; h.env_stride = env_stride; h.pix_stride = pix_stride;
    h.w1 = w.w1; h.b1 = w.b1; h.s1 = w.s1; h.t1 = w.t1; h.w2t = w.w2; h.b2 = w.b2; h.K1 = w.K1; h.NOUT = w.NOUT;
    h.categorical = categorical; h.support_min = support_min; h.out_logits = out_logits; h.out_scalar = out_scalar;
    return h;
}

// head debug buffers (lz_roots_enable_trace(r, 3)): every simulation's support-wide logits and pre-transform expectations, per pool slot
static int ensure_head_debug(lz_roots *r)
{
    if (!r->head_debug || r->hd_logits) return LZ_OK;
    const lz_model_cfg &c = r->eng->model->cfg;
    LZ_REQUIRE(c.model_type < 2, "head debug buffers exist for the conv models only");
    LZ_REQUIRE(!r->eng->model->wide_heads, "head debug buffers: not with the dense-layer heads of the conv Sampled EfficientZero (its support-wide logits: lz_roots_read_debug_logits)");
    const size_t B = r->t.B, NN = r->t.NN, SUP = std::max(c.support_size, c.reward_support_size);
    const size_t n = NN * 2 * B * (SUP + 1);
    hipError_t err = lz_dev_malloc((void **)&r->hd_logits, n * 4);
    if (err != hipSuccess) {
        r->hd_logits = nullptr;
        lz_set_error("hipMalloc(%zu bytes) for the head debug buffers failed: %s", n * 4, hipGetErrorString(err));
        return err == hipErrorOutOfMemory ? LZ_ERR_NOMEM : LZ_ERR_HIP;
    }
    r->hd_expect = r->hd_logits + NN * 2 * B * SUP;
    r->hd_sup = SUP;
    LZ_HIP_CHECK(hipMemsetAsync(r->hd_logits, 0, n * 4, r->eng->stream));
    return LZ_OK;
}
static float *hd_logits_at(lz_roots *r, int slot, int which) { return r->hd_logits + ((size_t)slot * 2 + which) * r->t.B * r->hd_sup; }
static float *hd_expect_at(lz_roots *r, int slot, int which) { return r->hd_expect + ((size_t)slot * 2 + which) * r->t.B; }

// The heads of the convolutional Sampled EfficientZero (hidden width up to 256, ReLU | GELU): dense layers + row finishers of the MLP family's
// kernels -- first layers (value | policy [| value prefix]) in one k_dense launch, second layers in another, then softmax . support -> h^-1.
// The 1x1 head convolutions wrote their outputs to two contiguous [B][HW * HC] halves of t_pv (chain_args_for / lz_initial_inference).
static void wide_heads(lz_roots *r, float *out_value, float *out_logits, bool with_vp, float *out_vp, hipStream_t s)
{
    lz_model *m = r->eng->model;
    const lz_model_cfg &c = m->cfg;
    const int B = r->t.B, HW = m->HWl, HC = c.head_channels;
    const int acts[3] = {act_pred(c), act_pred(c), act_dyn(c)};   // lz_dense_job.act: 1 ReLU, 2 GELU(tanh); the value-prefix head belongs to the dynamics network
    const WideHead *w[3] = {&m->wh_value, &m->wh_policy, &m->wh_reward};
    const float *x[3] = {r->t_pv, r->t_pv + (size_t)B * HW * HC, r->t_hbn};
    float *hid[3] = {r->t_x1, r->t_x2, r->t_x3};
    float *lg[3] = {r->dbg_logits[0], out_logits, r->dbg_logits[1]};
    const int n = with_vp ? 3 : 2;
    lz_dense_args a1{}, a2{};
    a1.B = a2.B = B;
    a1.njobs = a2.njobs = n;
    for (int i = 0; i < n; ++i) {
        lz_dense_job &j = a1.job[i];
        j.x = x[i]; j.K1 = w[i]->K1; j.wf = w[i]->w1f; j.bias = w[i]->b1; j.scale = w[i]->s1; j.shift = w[i]->t1; j.N = w[i]->HID; j.act = acts[i];
        j.out = hid[i];
        lz_dense_job &k = a2.job[i];
        k.x = hid[i]; k.K1 = w[i]->HID; k.wf = w[i]->w2f; k.bias = w[i]->b2; k.N = w[i]->NOUT; k.act = 0; k.out = lg[i];
    }
    lz_launch_dense(a1, s);
    lz_launch_dense(a2, s);
    lz_rowfinal_args f{};
    f.B = B;
    f.job[f.njobs++] = lz_rowfinal_job{r->dbg_logits[0], m->wh_value.NOUT, c.support_min, out_value};
    if (with_vp) f.job[f.njobs++] = lz_rowfinal_job{r->dbg_logits[1], m->wh_reward.NOUT, c.support_min, out_vp};
    lz_launch_rowfinal(f, s);
}

// the head MLPs (value, policy[, value prefix]) in one launch; inputs are the 1x1-conv outputs t_pv / the LSTM output.
// slot = the pool slot the outputs belong to (head debug buffers)
static void heads(lz_roots *r, int slot, float *out_value, float *out_logits, float *dbg_value_logits, bool with_vp, float *out_vp,
                  float *dbg_vp_logits, hipStream_t s)
{
    lz_model *m = r->eng->model;
    if (m->wide_heads) { wide_heads(r, out_value, out_logits, with_vp, out_vp, s); return; }
    const lz_model_cfg &c = m->cfg;
    const int B = r->t.B, HW = m->HWl, HC = c.head_channels;
    lz_head_desc h[3];
    int n = 0;
    // the support-wide logits are observability for the parity tests (lz_roots_read_debug_logits): written only while tracing is on
    if (!r->trace_on) dbg_value_logits = dbg_vp_logits = nullptr;
    const bool hd = r->trace_on && r->head_debug && r->hd_logits;   // per-slot copies (+ the pre-transform expectations)
    h[n++] = headdesc(m->fc_value, r->t_pv, HW * 2 * HC, 2 * HC, 1, c.support_min, dbg_value_logits, out_value);
    if (hd) h[n - 1].out_expect = hd_expect_at(r, slot, 0);
    h[n++] = headdesc(m->fc_policy, r->t_pv + HC, HW * 2 * HC, 2 * HC, 0, 0.f, out_logits, nullptr);
    if (with_vp) {
        if (c.model_type == 1) h[n++] = headdesc(m->fc_reward, r->t_rx, HW * HC, HC, 1, c.reward_support_size > 0 ? c.reward_support_min : c.support_min, dbg_vp_logits, out_vp);
        else h[n++] = headdesc(m->fc_reward, r->t_hbn, c.lstm_hidden_size, 16, 1, c.support_min, dbg_vp_logits, out_vp);
        if (hd) h[n - 1].out_expect = hd_expect_at(r, slot, 1);
    }
    lz_launch_heads(h, n, B, 32, s);  // Builder::mlp pads narrower heads to the compiled 32 hidden units
    if (hd) {   // the launch wrote its logits [B][NOUT] to the "latest head launch" buffers: keep a copy with the slot
        if (dbg_value_logits) (void)hipMemcpyAsync(hd_logits_at(r, slot, 0), dbg_value_logits, (size_t)B * m->fc_value.NOUT * 4, hipMemcpyDeviceToDevice, s);
        if (with_vp && dbg_vp_logits) (void)hipMemcpyAsync(hd_logits_at(r, slot, 1), dbg_vp_logits, (size_t)B * m->fc_reward.NOUT * 4, hipMemcpyDeviceToDevice, s);
    }
}

// records a HIP-event pair around one launch when in-stream profiling is on (bench.py roofline)
struct ProfScope {
    lz_engine *e;
    hipStream_t s;
    bool rec;
    ProfScope(lz_engine *e_, hipStream_t s_) : e(e_), s(s_)
    {
        rec = e && e->prof_on && 2 * (e->prof_used + 1) <= e->prof_ev.size();
        if (rec) (void)hipEventRecord(e->prof_ev[2 * e->prof_used], s);
    }
    ~ProfScope()
    {
        if (rec) { (void)hipEventRecord(e->prof_ev[2 * e->prof_used + 1], s); e->prof_used++; }
    }
};

extern "C" int lz_initial_inference(lz_roots *r, const float *d_obs)
{
    LZ_REQUIRE(r != nullptr && d_obs != nullptr, "NULL argument");
    lz_model *m = r->eng->model;
    if (!m || !m->finalized) { lz_set_error("no finalized model on this engine"); return LZ_ERR_STATE; }
    {
        const int mt = m->cfg.model_type;
        const bool conv_sampled = mt == 0 && m->cfg.num_of_sampled_actions > 0;   // SampledEfficientZeroModel (conv), discrete actions
        const int want = conv_sampled ? LZ_TREE_SAMPLED_EFFICIENTZERO : (mt == 0 || mt == 3) ? LZ_TREE_EFFICIENTZERO : (mt == 4 ? LZ_TREE_SAMPLED_EFFICIENTZERO : LZ_TREE_MUZERO);
        LZ_REQUIRE(r->t.variant == want || (want == LZ_TREE_MUZERO && r->t.variant == LZ_TREE_GUMBEL_MUZERO), "tree variant does not match the model type (EfficientZero model <-> EZ tree, MuZero model <-> MZ tree, sampled model <-> sampled tree)");
        if (mt == 4 || conv_sampled) {
            const bool cont = m->cfg.action_encoding == 2;
            LZ_REQUIRE(r->t.A == m->cfg.num_of_sampled_actions, "sampled roots: num_of_sampled_actions differs from the model's");
            LZ_REQUIRE(cont ? (r->t.disc_A == 0 && r->t.D == m->cfg.action_space_size) : (r->t.disc_A == m->cfg.action_space_size),
                       "sampled roots: action space (continuous dimension / discrete size) differs from the model's");
        }
        else LZ_REQUIRE(r->t.A == m->cfg.action_space_size, "roots action space differs from the model's");
    }
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    if (m->cfg.model_type >= 2) {
        const int rc = lz_mlp_initial_inference(r, d_obs);
        if (rc == LZ_OK) r->last_obs = d_obs;   // the env-step rows take their newest observation from here
        return rc;
    }
    int rc = ensure_pools(r);
    if (rc != LZ_OK) return rc;
    if (r->trace_on && r->head_debug && (rc = ensure_head_debug(r)) != LZ_OK) return rc;
    rc = ensure_ws(m, r->t.B);
    if (rc != LZ_OK) return rc;
    hipStream_t s = r->eng->stream;
    const lz_model_cfg &c = m->cfg;
    const int B = r->t.B, C = c.num_channels, H = c.model_type == 0 ? c.lstm_hidden_size : 0;
    float *w0 = m->ws[0], *w1 = m->ws[1], *w2 = m->ws[2];
    if (!c.downsample) {
        lz_launch_conv_in(d_obs, m->rin.w, m->rin.scale, m->rin.shift, w0, B, c.obs_c, c.obs_h, c.obs_w, C, s);
    } else {
    // DownSample (common.py:266-365)
    int stage = 0;
#define LZ_STAGE() do { if (m->debug_stop == ++stage) { LZ_HIP_CHECK(hipStreamSynchronize(s)); return LZ_OK; } } while (0)
    // grid sizes for 96 | 64 observations: S1 = 48 | 32, S2 = 24 | 16, S3 = 12 | 8, then 6 | (no second pooling)
    const int S1 = c.obs_h / 2, S2 = (S1 + 1) / 2, S3 = (S2 + 1) / 2;
    const int bf = c.precision == 1 ? 1 : 0;   // fast mode: the tower's activations are bf16 NHWC tensors (in the same workspaces), fp32 again out of the last pooling
    lz_launch_conv_first(d_obs, m->first_w, m->first_s, m->first_t, w0, B, c.obs_c, c.obs_h, c.obs_w, C / 2, s, bf);  // S1 x S1 x 32
    LZ_STAGE();
    conv(m->r1a, w0, w1, B, S1, S1, 1, nullptr, 1, s, bf);
    LZ_STAGE();
    conv(m->r1b, w1, w2, B, S1, S1, 1, w0, 1, s, bf);            // w2: S1 x S1 x 32
    LZ_STAGE();
    // conv1 (w0: S2 x S2 x 64) and the identity path's conv3 (w1: no norm, no act) read the same tensor: one launch where the pair has an instance
    const bool pair = m->debug_stop == 0 && m->dn1.cin == m->dn3.cin &&
                      lz_launch_conv3x3_pair(conv_args(m->dn1, w2, w0, B, S1, S2, nullptr, 1, bf), conv_args(m->dn3, w2, w1, B, S1, S2, nullptr, 0, bf), m->dn1.cin, 2, s);
    if (!pair) conv(m->dn1, w2, w0, B, S1, S2, 2, nullptr, 1, s, bf);
    LZ_STAGE();
    if (!pair) conv(m->dn3, w2, w1, B, S1, S2, 2, nullptr, 0, s, bf);
    LZ_STAGE();
    conv(m->dn2, w0, w2, B, S2, S2, 1, w1, 1, s, bf);            // w2: S2 x S2 x 64
    LZ_STAGE();
    conv(m->r2a, w2, w0, B, S2, S2, 1, nullptr, 1, s, bf);
    LZ_STAGE();
    conv(m->r2b, w0, w1, B, S2, S2, 1, w2, 1, s, bf);            // w1
    LZ_STAGE();
    lz_launch_avgpool(w1, w0, B, S2, S2, C, s, bf, bf);               // w0: S3 x S3 x 64
    LZ_STAGE();
    conv(m->r3a, w0, w1, B, S3, S3, 1, nullptr, 1, s, bf);
    LZ_STAGE();
    conv(m->r3b, w1, w2, B, S3, S3, 1, w0, 1, s, bf);            // w2
    LZ_STAGE();
    if (c.obs_h == 64) {                                      // common.py:358-359: no second pooling
        if (bf) lz_launch_bf16_to_f32(w2, w0, (size_t)B * S3 * S3 * C, s);
        else LZ_HIP_CHECK(hipMemcpyAsync(w0, w2, (size_t)B * S3 * S3 * C * 4, hipMemcpyDeviceToDevice, s));
    } else {
        lz_launch_avgpool(w2, w0, B, S3, S3, C, s, bf, 0);           // w0: 6x6x64
    }
    LZ_STAGE();
    }
#undef LZ_STAGE
    // RepresentationNetwork.resblocks (common.py:775-776) -> latent pool slot 0, then the prediction residual block
    // and the value / policy 1x1 convs, all in one LDS-resident chain launch
    {
        lz_chain_args ca{};
        ca.in = w0; ca.B = B; ca.gw = m->GW; ca.gh = m->GH; ca.C = C;
        const int x_lat = chain_blocks(ca, m->rep_res, 0, -1, r->latent_pool);
        const int x_p = chain_blocks(ca, m->pred_res, x_lat, -1, nullptr, act_pred(c));
        ca.c1[0] = c1job(m->val_c, nullptr, r->t_pv, 2 * c.head_channels, 0); ca.c1_in[0] = x_p;
        ca.c1[1] = c1job(m->pol_c, nullptr, r->t_pv, 2 * c.head_channels, c.head_channels); ca.c1_in[1] = x_p;
        ca.nc1 = 2;
        if (m->wide_heads) {   // dense-layer heads read two contiguous [B][HW * HC] blocks
            ca.c1[0] = c1job(m->val_c, nullptr, r->t_pv, c.head_channels, 0);
            ca.c1[1] = c1job(m->pol_c, nullptr, r->t_pv + (size_t)B * m->HWl * c.head_channels, c.head_channels, 0);
        }
        ca.c1[0].act = ca.c1[1].act = act_pred(c);
        ca.gelu = act_pred(c) == 2;
        lz_launch_chain(ca, s);
    }
    if (H > 0) {   // reward_hidden_state_roots = zeros (efficientzero_model.py:229-236): slot 0 of both pools, one launch
        const size_t n4 = B * H / 4;
        if ((B * H) % 4 == 0) {
            hipLaunchKernelGGL(k_zero2, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, reinterpret_cast<float4 *>(r->h_pool),
                               reinterpret_cast<float4 *>(r->c_pool), n4);
        } else {
            LZ_HIP_CHECK(hipMemsetAsync(r->h_pool, 0, (size_t)B * H * 4, s));
            LZ_HIP_CHECK(hipMemsetAsync(r->c_pool, 0, (size_t)B * H * 4, s));
        }
    }
    heads(r, 0, r->sim_value, r->sim_logits, r->dbg_logits[0], false, nullptr, nullptr, s);
    LZ_HIP_CHECK(hipGetLastError());
    r->inferred = true;
    r->inference_fresh = true;  // no prepare has consumed it yet (lz_roots_reset_keep_inference)
    r->last_obs = d_obs;
    return LZ_OK;
}

extern "C" int lz_initial_inference_host(lz_roots *r, const float *h_obs)
{
    LZ_REQUIRE(r != nullptr && h_obs != nullptr, "NULL argument");
    lz_model *m = r->eng->model;
    if (!m || !m->finalized) { lz_set_error("no finalized model on this engine"); return LZ_ERR_STATE; }
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    if (int rc = lz_roots_release_pools_if_stale(r)) return rc;
    const size_t n = (size_t)r->t.B * m->cfg.obs_c * m->cfg.obs_h * m->cfg.obs_w;
    if (r->d_obs_bytes < n * 4) {
        if (r->d_obs) {
            LZ_HIP_CHECK(hipStreamSynchronize(r->eng->stream));
            if (r->last_obs == r->d_obs) r->last_obs = nullptr;   // lz_roots_collect_rows must not read the freed staging buffer
            (void)hipFree(r->d_obs); r->d_obs = nullptr;
        }
        LZ_HIP_CHECK(lz_dev_malloc((void **)&r->d_obs, n * 4));
        r->d_obs_bytes = n * 4;
    }
    LZ_HIP_CHECK(hipMemcpyAsync(r->d_obs, h_obs, n * 4, hipMemcpyHostToDevice, r->eng->stream));
    return lz_initial_inference(r, r->d_obs);
}

// The reference's call order -- network_output = model.initial_inference(obs); roots = MCTSCtree.roots(...); roots.prepare(host
// lists); search(roots, model, latent_state_roots, ...) (efficientzero.py:582-610) -- infers BEFORE the roots exist.  The engine
// model then infers into a handle of its own (src) and the search adopts that inference into the prepared roots (dst): slot 0
// of the latent / LSTM pools and the root predictions move device to device, to_play of the host-side prepare is uploaded.
extern "C" int lz_roots_adopt_inference(lz_roots *dst, lz_roots *src)
{
    LZ_REQUIRE(dst != nullptr && src != nullptr && dst != src, "NULL / identical handles");
    LZ_REQUIRE(dst->eng == src->eng, "the two roots live on different engines");
    lz_model *m = dst->eng->model;
    LZ_REQUIRE(m != nullptr && m->finalized, "no finalized model on this engine");
    LZ_REQUIRE(src->inferred && src->pool_slab != nullptr && src->pool_model_uid == src->eng->model_uid, "the source roots hold no inference of the engine's current model");
    LZ_REQUIRE(dst->t.B == src->t.B && dst->t.variant == src->t.variant && dst->t.A == src->t.A && dst->t.D == src->t.D, "the two roots differ in shape");
    LZ_REQUIRE(dst->prepared && (int)dst->h_to_play.size() == dst->t.B, "adopt after a host-side Roots.prepare on the destination");
    LZ_HIP_CHECK(hipSetDevice(dst->eng->device));
    int rc = m->cfg.model_type >= 2 ? lz_mlp_ensure_pools(dst) : ensure_pools(dst);
    if (rc != LZ_OK) return rc;
    const size_t B = dst->t.B, C = m->cfg.num_channels, HW = m->HWl;
    const size_t H = m->cfg.model_type >= 2 ? (size_t)lz_mlp_hidden_size(m) : (size_t)(m->cfg.model_type == 0 ? m->cfg.lstm_hidden_size : 0);
    const size_t PA = policy_width_of(m, dst->t);
    hipStream_t s = dst->eng->stream;
    LZ_HIP_CHECK(hipMemcpyAsync(dst->latent_pool, src->latent_pool, B * HW * C * 4, hipMemcpyDeviceToDevice, s));
    if (H) {
        LZ_HIP_CHECK(hipMemcpyAsync(dst->h_pool, src->h_pool, B * H * 4, hipMemcpyDeviceToDevice, s));
        LZ_HIP_CHECK(hipMemcpyAsync(dst->c_pool, src->c_pool, B * H * 4, hipMemcpyDeviceToDevice, s));
    }
    LZ_HIP_CHECK(hipMemcpyAsync(dst->sim_value, src->sim_value, B * 4, hipMemcpyDeviceToDevice, s));
    LZ_HIP_CHECK(hipMemcpyAsync(dst->sim_vp, src->sim_vp, B * 4, hipMemcpyDeviceToDevice, s));
    LZ_HIP_CHECK(hipMemcpyAsync(dst->sim_logits, src->sim_logits, B * PA * 4, hipMemcpyDeviceToDevice, s));
    LZ_HIP_CHECK(hipMemcpyAsync(dst->d_to_play, dst->h_to_play.data(), B * 4, hipMemcpyHostToDevice, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));   // h_to_play is pageable host memory
    dst->inferred = true;
    dst->inference_fresh = false;
    dst->last_obs = src->last_obs;
    return LZ_OK;
}

extern "C" int lz_roots_get_root_outputs(lz_roots *r, float *h_pred_values, float *h_policy_logits)
{
    LZ_REQUIRE(r != nullptr && r->inferred, "lz_initial_inference has not run on these roots");
    const size_t B = r->t.B, A = policy_width_of(r->eng->model, r->t);
    hipStream_t s = r->eng->stream;
    if (h_pred_values) LZ_HIP_CHECK(hipMemcpyAsync(h_pred_values, r->sim_value, B * 4, hipMemcpyDeviceToHost, s));
    if (h_policy_logits) LZ_HIP_CHECK(hipMemcpyAsync(h_policy_logits, r->sim_logits, B * A * 4, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    return LZ_OK;
}

// everything _forward_collect reads back after a search, in ONE readout launch, one device-to-host copy and one
// synchronisation: visit-count distributions (+ counts), root values, and the root predictions of lz_initial_inference
static int search_results(lz_roots *r, int32_t *h_out_dist, int32_t *h_out_count, float *h_out_values, float *h_pred_values,
                          float *h_policy_logits, bool select, double temperature, int deterministic, uint64_t seed,
                          int32_t *h_action_pos, double *h_entropy)
{
    LZ_REQUIRE(r != nullptr && h_out_dist != nullptr && h_out_count != nullptr && h_out_values != nullptr, "NULL argument");
    LZ_REQUIRE(r->prepared && r->inferred && r->pool_slab != nullptr, "roots not searched through the fused path");
    LZ_REQUIRE(r->t.variant != LZ_TREE_SAMPLED_EFFICIENTZERO, "use the lz_sroots_* getters for sampled roots");
    LZ_REQUIRE(!select || (temperature > 0.0 && h_action_pos != nullptr && h_entropy != nullptr), "select_action needs a positive temperature and output arrays");
    const lz_tree_dev &t = r->t;
    const size_t B = t.B, A = t.A;
    const size_t PA = policy_width_of(r->eng->model, r->t);
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    hipStream_t s = r->eng->stream;
    // layout of the result block: entropy [B] f64 | dist [B][A] | count [B] | action pos [B] | values [B] | pred values [B] | logits [B][PA]
    const size_t n_i = B * A + 2 * B, n_f = 2 * B + B * PA, bytes = B * 8 + (n_i + n_f) * 4;
    if (r->results_bytes < bytes) {
        if (r->d_results) { LZ_HIP_CHECK(hipStreamSynchronize(s)); (void)hipFree(r->d_results); r->d_results = nullptr; }
        if (r->h_results) { (void)hipHostFree(r->h_results); r->h_results = nullptr; }
        LZ_HIP_CHECK(lz_dev_malloc((void **)&r->d_results, bytes));
        LZ_HIP_CHECK(hipHostMalloc(&r->h_results, bytes, hipHostMallocDefault));
        r->results_bytes = bytes;
    }
    double *d_ent = (double *)r->d_results;
    int32_t *d_dist = (int32_t *)(d_ent + B), *d_cnt = d_dist + B * A, *d_pos = d_cnt + B;
    float *d_val = (float *)(d_pos + B), *d_pred = d_val + B, *d_lg = d_pred + B;
    lz_tree_launch_readout(t, d_dist, d_cnt, d_val, s);
    if (select) lz_launch_select_action(t, 1.0 / temperature, deterministic, seed, d_pos, d_ent, s);
    LZ_HIP_CHECK(hipGetLastError());
    if (h_pred_values) LZ_HIP_CHECK(hipMemcpyAsync(d_pred, r->sim_value, B * 4, hipMemcpyDeviceToDevice, s));
    if (h_policy_logits) LZ_HIP_CHECK(hipMemcpyAsync(d_lg, r->sim_logits, B * PA * 4, hipMemcpyDeviceToDevice, s));
    LZ_HIP_CHECK(hipMemcpyAsync(r->h_results, r->d_results, bytes, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    const double *he = (const double *)r->h_results;
    const int32_t *hi = (const int32_t *)(he + B);
    const float *hf = (const float *)(hi + n_i);
    memcpy(h_out_dist, hi, B * A * 4);
    memcpy(h_out_count, hi + B * A, B * 4);
    memcpy(h_out_values, hf, B * 4);
    if (h_pred_values) memcpy(h_pred_values, hf + B, B * 4);
    if (h_policy_logits) memcpy(h_policy_logits, hf + 2 * B, B * PA * 4);
    if (select) {
        memcpy(h_action_pos, hi + B * A + B, B * 4);
        memcpy(h_entropy, he, B * 8);
    }
    return LZ_OK;
}

extern "C" int lz_roots_get_search_results(lz_roots *r, int32_t *h_out_dist, int32_t *h_out_count, float *h_out_values,
                                           float *h_pred_values, float *h_policy_logits)
{
    return search_results(r, h_out_dist, h_out_count, h_out_values, h_pred_values, h_policy_logits, false, 1.0, 1, 0, nullptr, nullptr);
}

extern "C" int lz_roots_get_search_results_select(lz_roots *r, int32_t *h_out_dist, int32_t *h_out_count, float *h_out_values,
                                                  float *h_pred_values, float *h_policy_logits, double temperature,
                                                  int deterministic, uint64_t seed, int32_t *h_action_pos, double *h_entropy)
{
    return search_results(r, h_out_dist, h_out_count, h_out_values, h_pred_values, h_policy_logits, true, temperature, deterministic,
                          seed, h_action_pos, h_entropy);
}

// ------------------------------------------------------------------------------------------------
// Env-step rows: what MuZeroCollector keeps per env and step (muzero_collector.py:557-620) in the field set of
// GameSegment.append / store_search_stats (game_segment.py:158-182, 241-263), written by ONE kernel from device buffers --
// the packed trajectory row of SURVEY.md 8(e)/(f4), the unit of the RCCL all-gather.  float32 words:
//   0 action (in the full action space)   1 reward (0 here: the environment fills it after stepping)   2 root value (searched)
//   3 predicted value   4 to_play   5 timestep   6 visit-count-distribution entropy (bits)   7 number of legal actions
//   8 .. 8+A      child visits / sum of visits in LEGAL-LIST order, zero padded (store_search_stats, game_segment.py:247-252)
//   8+A .. 8+2A   action mask (1 legal / 0 not)
//   8+2A ..       the newest observation frame: the last `frame_floats` floats of the env's observation (image_channel x H x W of
//                 the stacked [C, H, W] input; the whole vector for vector observations) -- what obs_segment holds for the step
// ------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ uint64_t sel_mix64c(uint64_t z)   // the generator of k_select_action (lz_capi.hip): same seed, same draw
{
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void k_pack_rows(lz_tree_dev t, const int32_t *__restrict__ dist, const int32_t *__restrict__ cnt,
                                                   const float *__restrict__ values, const float *__restrict__ pred,
                                                   const int32_t *__restrict__ pos, const double *__restrict__ ent,
                                                   const int32_t *__restrict__ to_play, const int32_t *__restrict__ timestep,
                                                   const float *__restrict__ obs, int obs_floats, int frame_floats,
                                                   float *__restrict__ rows, int row_words)
{
    const int b = blockIdx.x, tid = threadIdx.x, A = t.A;
    float *row = rows + (size_t)b * row_words;
    const int n = cnt[b];
    if (tid == 0) {
        int total = 0;
        for (int j = 0; j < n; ++j) total += dist[(size_t)b * A + j];
        const float sum_visits = total == 0 ? 1e-6f : (float)total;  // game_segment.py:244-246
        row[0] = (float)t.legal[(size_t)b * A + pos[b]];
        row[1] = 0.0f;
        row[2] = values[b];
        row[3] = pred[b];
        row[4] = (float)to_play[b];
        row[5] = timestep ? (float)timestep[b] : -1.0f;
        row[6] = (float)ent[b];
        row[7] = (float)n;
        for (int j = 0; j < A; ++j) row[8 + j] = j < n ? (float)dist[(size_t)b * A + j] / sum_visits : 0.0f;
        for (int j = 0; j < A; ++j) row[8 + A + j] = 0.0f;
        for (int j = 0; j < n; ++j) row[8 + A + t.legal[(size_t)b * A + j]] = 1.0f;
    }
    if (frame_floats > 0) {
        const float *src = obs + (size_t)b * obs_floats + (obs_floats - frame_floats);
        float *dst = row + 8 + 2 * A;
        for (int i = tid; i < frame_floats; i += 256) dst[i] = src[i];
    }
}
// The same row, with everything before it, in ONE launch (A <= 64): readout of the root (get_distributions / get_values,
// cnode.cpp:389-419), select_action (lzero/policy/utils.py:637-661, float64 like k_select_action and with the same draw) on the lanes of
// wave 0 -- lane j owns legal action j, the order-sensitive sums (normaliser, cumulative distribution, entropy) run in list order over
// v_readlane -- and the row.  The header words and the root's policy logits are ALSO written straight into pinned host memory
// (h_header / h_logits are device-visible pointers), the time steps are read from it: the read-back is this kernel plus one
// synchronisation instead of three kernels, four copies and the gaps between them (~35 us per collect step).
__device__ __forceinline__ double rl_d(double v, int lane)
{
    const long long x = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(x & 0xffffffffll), lane), hi = __builtin_amdgcn_readlane((int)(x >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__global__ __launch_bounds__(256) void k_collect_rows(lz_tree_dev t, double inv_temperature, int deterministic, uint64_t seed,
                                                      const float *__restrict__ pred, const float *__restrict__ root_logits, int PA,
                                                      const int32_t *__restrict__ to_play, const int32_t *__restrict__ timestep,
                                                      const float *__restrict__ obs, int obs_floats, int frame_floats,
                                                      float *__restrict__ rows, int row_words, float *__restrict__ h_header,
                                                      float *__restrict__ h_logits)
{
    const int b = blockIdx.x, tid = threadIdx.x, A = t.A, NN = t.NN;
    float *row = rows + (size_t)b * row_words;
    if (tid < 64) {
        const int lane = tid, n = t.n_legal[b];
        const bool live = lane < n;
        const int act = live ? t.legal[(size_t)b * A + lane] : 0;
        const int c = live ? __float_as_int(t.edge[((size_t)b * NN) * A + act].y) : -1;
        // N^(1/T); T = 1 (the collector's setting for most of training) needs no pow: x^1 is x
        const double pw = live ? (inv_temperature == 1.0 ? (double)c : pow((double)c, inv_temperature)) : 0.0;
        double sum = 0.0;
        int total = 0, best = -1, arg = 0;
        for (int j = 0; j < n; ++j) {
            sum += rl_d(pw, j);
            const int cj = __builtin_amdgcn_readlane(c, j);
            total += cj;
            if (cj > best) { best = cj; arg = j; }  // np.argmax: first maximum
        }
        const double p = live ? pw / sum : 0.0;
        const double plog = (live && p > 0.0) ? p * log2(p) : 0.0;
        const double u = (double)(sel_mix64c(sel_mix64c(seed) ^ (uint64_t)b) >> 11) * (1.0 / 9007199254740992.0);  // [0, 1)
        double acc = 0.0, H = 0.0;
        int pick = -1, last = 0;
        for (int j = 0; j < n; ++j) {
            const double pj = rl_d(p, j);
            if (pj > 0.0) { H -= rl_d(plog, j); last = j; }
            acc += pj;
            if (pick < 0 && u < acc) pick = j;  // searchsorted(cumsum(p), u, side='right') like np.random.choice
        }
        if (pick < 0) pick = last;
        const int pos = deterministic ? arg : pick;
        const int rv = t.root_visit[b];
        const float value = (rv == 0) ? 0.0f : t.root_vsum[b] / (float)rv;
        const float sum_visits = total == 0 ? 1e-6f : (float)total;  // game_segment.py:244-246
        const int hw = 8 + 2 * A;
        float *hh = h_header + (size_t)b * hw;
        // header words: lane 0 the eight scalars, lanes < A the visit share of list position `lane` and the mask bit of action `lane`
        if (lane == 0) {
            const float w0 = (float)__builtin_amdgcn_readlane(act, pos), w4 = (float)to_play[b], w5 = timestep ? (float)timestep[b] : -1.0f;
            const float w3 = pred[b], w6 = (float)H, w7 = (float)n;
            row[0] = w0; row[1] = 0.0f; row[2] = value; row[3] = w3; row[4] = w4; row[5] = w5; row[6] = w6; row[7] = w7;
            hh[0] = w0; hh[1] = 0.0f; hh[2] = value; hh[3] = w3; hh[4] = w4; hh[5] = w5; hh[6] = w6; hh[7] = w7;
        }
        if (lane < A) {
            const float share = live ? (float)c / sum_visits : 0.0f;
            row[8 + lane] = share;
            hh[8 + lane] = share;
            // mask[a] = 1 iff a is in the legal list: lane a scans the list (n <= A <= 64)
            float mk = 0.0f;
            for (int j = 0; j < n; ++j) if (__builtin_amdgcn_readlane(act, j) == lane) mk = 1.0f;
            row[8 + A + lane] = mk;
            hh[8 + A + lane] = mk;
        }
        if (h_logits && lane < PA) h_logits[(size_t)b * PA + lane] = root_logits[(size_t)b * PA + lane];
    } else if (h_logits && PA > 64) {
        for (int i = tid; i < PA; i += 256) if (i >= 64) h_logits[(size_t)b * PA + i] = root_logits[(size_t)b * PA + i];
    }
    if (frame_floats > 0) {
        const float *src = obs + (size_t)b * obs_floats + (obs_floats - frame_floats);
        float *dst = row + 8 + 2 * A;
        if (((frame_floats | obs_floats | row_words | (2 * A)) & 3) == 0 && (((uintptr_t)obs | (uintptr_t)rows) & 15) == 0) {   // 16-byte aligned throughout
            const float4 *s4 = reinterpret_cast<const float4 *>(src);
            float4 *d4 = reinterpret_cast<float4 *>(dst);
            for (int i = tid; i < frame_floats / 4; i += 256) d4[i] = s4[i];
        } else {
            for (int i = tid; i < frame_floats; i += 256) dst[i] = src[i];
        }
    }
}
}  // namespace

extern "C" int lz_rows_width(int action_space_size, int frame_floats) { return 8 + 2 * action_space_size + frame_floats; }

// The weight transform of the Winograd kernels in a plain layout (host only, no device needed): u[p][ci][co] = (G g G^T)[p / 4][p % 4]
// of filter w[co][ci][3][3] -- what Builder::wino / wino_chain permute into fragment order.  tests/test_wino_cpu.py checks it (and
// the transform matrices the kernels hard-code) against an independent NumPy statement of F(2x2, 3x3).
extern "C" int lz_wino_weights(const float *w, int cout, int cin, float *u)
{
    LZ_REQUIRE(w != nullptr && u != nullptr && cout > 0 && cin > 0, "invalid argument");
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            double U[4][4];
            Builder::wino_u(w + ((size_t)co * cin + ci) * 9, U);
            for (int p = 0; p < 16; ++p) u[((size_t)p * cin + ci) * cout + co] = (float)U[p / 4][p % 4];
        }
    return LZ_OK;
}

// The read-back of the env-step rows is split in two so that a caller can keep the device busy: *_enqueue launches select_action + the
// row packing behind the search and records an event; collect_rows_finish waits for THAT event (not for the stream: another roots
// handle's search may already be queued behind it on the same engine -- the vectorised collector's env groups) and hands the header
// words over.  lz_roots_collect_rows[_ex] = enqueue + finish.
static int collect_rows_finish(lz_roots *r, float *h_header, float *h_policy_logits)
{
    LZ_REQUIRE(r != nullptr && h_header != nullptr, "NULL argument");
    LZ_REQUIRE(r->rows_pending, "no env-step rows in flight: call lz_roots_collect_rows_begin first");
    LZ_HIP_CHECK(hipEventSynchronize(r->rows_done));
    r->rows_pending = false;
    memcpy(h_header, r->rows_hh, r->rows_B * r->rows_hw * 4);
    if (h_policy_logits) {
        LZ_REQUIRE(r->rows_logits, "the rows were enqueued without the policy logits");
        memcpy(h_policy_logits, r->rows_hh + r->rows_B * r->rows_hw, r->rows_B * r->rows_pa * 4);
    }
    return LZ_OK;
}
static int collect_rows_mark(lz_roots *r, float *hh, size_t B, size_t hw, size_t pa, bool logits, hipStream_t s)
{
    if (!r->rows_done) LZ_HIP_CHECK(hipEventCreateWithFlags(&r->rows_done, hipEventDisableTiming));
    LZ_HIP_CHECK(hipEventRecord(r->rows_done, s));
    r->rows_hh = hh; r->rows_B = B; r->rows_hw = hw; r->rows_pa = pa; r->rows_logits = logits; r->rows_pending = true;
    return LZ_OK;
}

static int collect_rows_enqueue(lz_roots *r, double temperature, int deterministic, uint64_t seed, const float *d_obs,
                                int frame_floats, const int32_t *h_timestep, float *d_rows, int row_words, bool want_logits)
{
    LZ_REQUIRE(r != nullptr && d_rows != nullptr, "NULL argument");
    LZ_REQUIRE(!r->rows_pending, "env-step rows already in flight: call lz_roots_collect_rows_end first");
    float *h_policy_logits = want_logits ? (float *)1 : nullptr;   // (only its null-ness is used below)
    LZ_REQUIRE(r->prepared && r->inferred && r->pool_slab != nullptr, "roots not searched through the fused path");
    LZ_REQUIRE(r->t.variant != LZ_TREE_SAMPLED_EFFICIENTZERO, "sampled roots: use lz_roots_get_search_results + the lz_sroots_* getters");
    LZ_REQUIRE(temperature > 0.0, "select_action needs a positive temperature");
    const lz_tree_dev &t = r->t;
    lz_model *m = r->eng->model;
    const size_t B = t.B, A = t.A;
    const size_t PA = policy_width_of(m, r->t);
    const int obs_floats = m->cfg.obs_c * m->cfg.obs_h * m->cfg.obs_w;
    if (!d_obs) d_obs = r->last_obs;
    LZ_REQUIRE(frame_floats >= 0 && frame_floats <= obs_floats && (frame_floats == 0 || d_obs != nullptr), "frame_floats exceeds the observation / no observation known");
    LZ_REQUIRE(row_words >= (int)(8 + 2 * A) + frame_floats, "row_words too small (lz_rows_width)");
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    hipStream_t s = r->eng->stream;
    const size_t hw = 8 + 2 * A;
    const size_t n_i = B * A + 2 * B + B, n_f = B + B * PA, dev_bytes = B * 8 + (n_i + n_f) * 4;  // + timestep [B]
    const size_t ts_bytes = align_up(B * 4, 4096), host_bytes = ts_bytes + (B * hw + B * PA) * 4;   // pinned: timestep | headers | logits
    const size_t bytes = dev_bytes > host_bytes ? dev_bytes : host_bytes;
    if (r->results_bytes < bytes) {
        if (r->d_results) { LZ_HIP_CHECK(hipStreamSynchronize(s)); (void)hipFree(r->d_results); r->d_results = nullptr; }
        if (r->h_results) { (void)hipHostFree(r->h_results); r->h_results = nullptr; }
        LZ_HIP_CHECK(lz_dev_malloc((void **)&r->d_results, bytes));
        LZ_HIP_CHECK(hipHostMalloc(&r->h_results, bytes, hipHostMallocDefault));
        r->results_bytes = bytes;
    }
    float *hh = (float *)((char *)r->h_results + ts_bytes);
    if (A <= 64) {   // one launch; headers / logits land in the pinned block, the time steps are read from it
        if (h_timestep) memcpy(r->h_results, h_timestep, B * 4);
        hipLaunchKernelGGL(k_collect_rows, dim3((unsigned)B), dim3(256), 0, s, t, 1.0 / temperature, deterministic, seed, r->sim_value,
                           r->sim_logits, (int)PA, r->d_to_play, h_timestep ? (const int32_t *)r->h_results : nullptr, d_obs, obs_floats,
                           frame_floats, d_rows, row_words, hh, h_policy_logits ? hh + B * hw : nullptr);
        LZ_HIP_CHECK(hipGetLastError());
        return collect_rows_mark(r, hh, B, hw, PA, want_logits, s);
    }
    double *d_ent = (double *)r->d_results;
    int32_t *d_dist = (int32_t *)(d_ent + B), *d_cnt = d_dist + B * A, *d_pos = d_cnt + B, *d_ts = d_pos + B;
    float *d_val = (float *)(d_ts + B), *d_lg = d_val + B;
    lz_tree_launch_readout(t, d_dist, d_cnt, d_val, s);
    lz_launch_select_action(t, 1.0 / temperature, deterministic, seed, d_pos, d_ent, s);
    if (h_timestep) {
        memcpy(r->h_results, h_timestep, B * 4);
        LZ_HIP_CHECK(hipMemcpyAsync(d_ts, r->h_results, B * 4, hipMemcpyHostToDevice, s));
    }
    hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)B), dim3(256), 0, s, t, d_dist, d_cnt, d_val, r->sim_value, d_pos, d_ent,
                       r->d_to_play, h_timestep ? d_ts : nullptr, d_obs, obs_floats, frame_floats, d_rows, row_words);
    LZ_HIP_CHECK(hipGetLastError());
    // the header words of every row (what the collector needs to step the environments) and the root policy logits come back
    // in one synchronisation; the frames stay in HBM for the all-gather
    LZ_HIP_CHECK(hipMemcpy2DAsync(hh, hw * 4, d_rows, (size_t)row_words * 4, hw * 4, B, hipMemcpyDeviceToHost, s));
    if (h_policy_logits) {
        LZ_HIP_CHECK(hipMemcpyAsync(d_lg, r->sim_logits, B * PA * 4, hipMemcpyDeviceToDevice, s));
        LZ_HIP_CHECK(hipMemcpyAsync(hh + B * hw, d_lg, B * PA * 4, hipMemcpyDeviceToHost, s));
    }
    return collect_rows_mark(r, hh, B, hw, PA, want_logits, s);
}

extern "C" int lz_roots_collect_rows(lz_roots *r, double temperature, int deterministic, uint64_t seed, const float *d_obs,
                                     int frame_floats, const int32_t *h_timestep, float *d_rows, int row_words,
                                     float *h_header, float *h_policy_logits)
{
    LZ_REQUIRE(h_header != nullptr, "NULL argument");
    if (int rc = collect_rows_enqueue(r, temperature, deterministic, seed, d_obs, frame_floats, h_timestep, d_rows, row_words, h_policy_logits != nullptr)) return rc;
    return collect_rows_finish(r, h_header, h_policy_logits);
}

// ---- env-step rows of the two further families that ship: an EXTRA block between the action mask and the frame carries what
// their store_search_stats keeps besides visits and value (game_segment.py:254-258, muzero_collector.py:606-612):
//   Sampled EfficientZero   root_sampled_actions [K][D]; the visit block / mask / n_legal are over the K sampled actions (mask = 1),
//                           word 0 is the selected POSITION (the action is extra[pos D .. pos D + D))
//   Gumbel MuZero           improved_policy_probs [A] (softmax(logits + sigma(completed Q)), CRoots::get_policies cnode.cpp:506-541);
//                           word 0 is arg-max of the improved policy over the legal actions (gumbel_muzero.py:591-592)
namespace {
__global__ __launch_bounds__(256) void k_pack_rows_ex(lz_tree_dev t, int variant, const int32_t *__restrict__ dist, const int32_t *__restrict__ cnt,
                                                      const float *__restrict__ values, const float *__restrict__ pred,
                                                      const int32_t *__restrict__ pos, const double *__restrict__ ent,
                                                      const int32_t *__restrict__ to_play, const int32_t *__restrict__ timestep,
                                                      const float *__restrict__ extra, int extra_words,
                                                      const float *__restrict__ obs, int obs_floats, int frame_floats,
                                                      float *__restrict__ rows, int row_words)
{
    const int b = blockIdx.x, tid = threadIdx.x, A = t.A;
    float *row = rows + (size_t)b * row_words;
    const bool sampled = variant == LZ_TREE_SAMPLED_EFFICIENTZERO;
    const int n = sampled ? A : cnt[b];
    const float *ex = extra + (size_t)b * extra_words;
    if (tid == 0) {
        int total = 0;
        for (int j = 0; j < n; ++j) total += dist[(size_t)b * A + j];
        const float sum_visits = total == 0 ? 1e-6f : (float)total;  // game_segment.py:244-246
        float action = sampled ? (float)pos[b] : (float)t.legal[(size_t)b * A + pos[b]];
        if (variant == LZ_TREE_GUMBEL_MUZERO) {   // np.argmax over where(mask, improved, 0): first maximum, all-zero -> action 0
            float best = 0.0f;
            int arg = 0;
            for (int j = 0; j < n; ++j) {
                const int a = t.legal[(size_t)b * A + j];
                const float v = ex[a];
                if (v > best || (v == best && a < arg && v > 0.0f)) { best = v; arg = a; }
            }
            action = (float)arg;
        }
        row[0] = action;
        row[1] = 0.0f;
        row[2] = values[b];
        row[3] = pred[b];
        row[4] = (float)to_play[b];
        row[5] = timestep ? (float)timestep[b] : -1.0f;
        row[6] = (float)ent[b];
        row[7] = (float)n;
        for (int j = 0; j < A; ++j) row[8 + j] = j < n ? (float)dist[(size_t)b * A + j] / sum_visits : 0.0f;
        for (int j = 0; j < A; ++j) row[8 + A + j] = sampled ? 1.0f : 0.0f;
        if (!sampled) for (int j = 0; j < n; ++j) row[8 + A + t.legal[(size_t)b * A + j]] = 1.0f;
    }
    for (int i = tid; i < extra_words; i += 256) row[8 + 2 * A + i] = ex[i];
    if (frame_floats > 0) {
        const float *src = obs + (size_t)b * obs_floats + (obs_floats - frame_floats);
        float *dst = row + 8 + 2 * A + extra_words;
        for (int i = tid; i < frame_floats; i += 256) dst[i] = src[i];
    }
}
}  // namespace

extern "C" int lz_rows_extra_words(lz_roots *r)
{
    if (!r) return LZ_ERR_INVALID;
    if (r->t.variant == LZ_TREE_SAMPLED_EFFICIENTZERO) return r->t.A * r->t.D;
    if (r->t.variant == LZ_TREE_GUMBEL_MUZERO) return r->t.A;
    return 0;
}

static int collect_rows_enqueue_ex(lz_roots *r, double temperature, int deterministic, uint64_t seed, float discount_factor,
                                   const float *d_obs, int frame_floats, const int32_t *h_timestep, float *d_rows, int row_words, bool want_logits)
{
    LZ_REQUIRE(r != nullptr && d_rows != nullptr, "NULL argument");
    const int E = lz_rows_extra_words(r);
    if (E == 0) return collect_rows_enqueue(r, temperature, deterministic, seed, d_obs, frame_floats, h_timestep, d_rows, row_words, want_logits);
    LZ_REQUIRE(!r->rows_pending, "env-step rows already in flight: call lz_roots_collect_rows_end first");
    float *h_policy_logits = want_logits ? (float *)1 : nullptr;   // (only its null-ness is used below)
    LZ_REQUIRE(r->prepared && r->inferred && r->pool_slab != nullptr, "roots not searched through the fused path");
    LZ_REQUIRE(temperature > 0.0, "select_action needs a positive temperature");
    const lz_tree_dev &t = r->t;
    lz_model *m = r->eng->model;
    const bool sampled = t.variant == LZ_TREE_SAMPLED_EFFICIENTZERO;
    const size_t B = t.B, A = t.A;
    const size_t PA = policy_width_of(m, r->t);
    const int obs_floats = m->cfg.obs_c * m->cfg.obs_h * m->cfg.obs_w;
    if (!d_obs) d_obs = r->last_obs;
    LZ_REQUIRE(frame_floats >= 0 && frame_floats <= obs_floats && (frame_floats == 0 || d_obs != nullptr), "frame_floats exceeds the observation / no observation known");
    LZ_REQUIRE(row_words >= (int)(8 + 2 * A) + E + frame_floats, "row_words too small (lz_rows_width + lz_rows_extra_words)");
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    hipStream_t s = r->eng->stream;
    const size_t hw = 8 + 2 * A + (size_t)E;    // the header that returns to the host includes the extra block
    const size_t n_i = B * A + 3 * B, n_f = B + B * PA + B * (size_t)E, dev_bytes = B * 8 + (n_i + n_f) * 4;
    const size_t host_bytes = (B + B * hw + B * PA) * 4;
    const size_t bytes = dev_bytes > host_bytes ? dev_bytes : host_bytes;
    if (r->results_bytes < bytes) {
        if (r->d_results) { LZ_HIP_CHECK(hipStreamSynchronize(s)); (void)hipFree(r->d_results); r->d_results = nullptr; }
        if (r->h_results) { (void)hipHostFree(r->h_results); r->h_results = nullptr; }
        LZ_HIP_CHECK(lz_dev_malloc((void **)&r->d_results, bytes));
        LZ_HIP_CHECK(hipHostMalloc(&r->h_results, bytes, hipHostMallocDefault));
        r->results_bytes = bytes;
    }
    double *d_ent = (double *)r->d_results;
    int32_t *d_dist = (int32_t *)(d_ent + B), *d_cnt = d_dist + B * A, *d_pos = d_cnt + B, *d_ts = d_pos + B;
    float *d_val = (float *)(d_ts + B), *d_lg = d_val + B, *d_extra = d_lg + B * PA;
    if (sampled) {
        lz_stree_launch_readout(t, d_dist, d_val, s);
        // the roots' K sampled actions: node 0 of every root, [B] strided blocks of K * D floats
        LZ_HIP_CHECK(hipMemcpy2DAsync(d_extra, (size_t)E * 4, t.actions, (size_t)t.NN * E * 4, (size_t)E * 4, B, hipMemcpyDeviceToDevice, s));
    } else {
        lz_tree_launch_readout(t, d_dist, d_cnt, d_val, s);
        lz_gtree_launch_policies(t, discount_factor, d_extra, nullptr, s);
    }
    lz_launch_select_action(t, 1.0 / temperature, deterministic, seed, d_pos, d_ent, s);
    int32_t *hts = (int32_t *)r->h_results;
    if (h_timestep) {
        memcpy(hts, h_timestep, B * 4);
        LZ_HIP_CHECK(hipMemcpyAsync(d_ts, hts, B * 4, hipMemcpyHostToDevice, s));
    }
    hipLaunchKernelGGL(k_pack_rows_ex, dim3((unsigned)B), dim3(256), 0, s, t, (int)t.variant, d_dist, d_cnt, d_val, r->sim_value, d_pos, d_ent,
                       r->d_to_play, h_timestep ? d_ts : nullptr, d_extra, E, d_obs, obs_floats, frame_floats, d_rows, row_words);
    LZ_HIP_CHECK(hipGetLastError());
    float *hh = (float *)(hts + B);
    LZ_HIP_CHECK(hipMemcpy2DAsync(hh, hw * 4, d_rows, (size_t)row_words * 4, hw * 4, B, hipMemcpyDeviceToHost, s));
    if (h_policy_logits) {
        LZ_HIP_CHECK(hipMemcpyAsync(d_lg, r->sim_logits, B * PA * 4, hipMemcpyDeviceToDevice, s));
        LZ_HIP_CHECK(hipMemcpyAsync(hh + B * hw, d_lg, B * PA * 4, hipMemcpyDeviceToHost, s));
    }
    return collect_rows_mark(r, hh, B, hw, PA, want_logits, s);
}

extern "C" int lz_roots_collect_rows_ex(lz_roots *r, double temperature, int deterministic, uint64_t seed, float discount_factor,
                                        const float *d_obs, int frame_floats, const int32_t *h_timestep, float *d_rows, int row_words,
                                        float *h_header, float *h_policy_logits)
{
    LZ_REQUIRE(h_header != nullptr, "NULL argument");
    if (int rc = collect_rows_enqueue_ex(r, temperature, deterministic, seed, discount_factor, d_obs, frame_floats, h_timestep, d_rows, row_words, h_policy_logits != nullptr)) return rc;
    return collect_rows_finish(r, h_header, h_policy_logits);
}

extern "C" int lz_roots_collect_rows_begin(lz_roots *r, double temperature, int deterministic, uint64_t seed, float discount_factor,
                                           const float *d_obs, int frame_floats, const int32_t *h_timestep, float *d_rows, int row_words, int want_logits)
{
    return collect_rows_enqueue_ex(r, temperature, deterministic, seed, discount_factor, d_obs, frame_floats, h_timestep, d_rows, row_words, want_logits != 0);
}

extern "C" int lz_roots_collect_rows_end(lz_roots *r, float *h_header, float *h_policy_logits)
{
    return collect_rows_finish(r, h_header, h_policy_logits);
}

extern "C" int lz_sroots_set_given(lz_roots *r, const float *h_draws, int records)
{
    LZ_REQUIRE(r != nullptr && r->t.variant == LZ_TREE_SAMPLED_EFFICIENTZERO, "not a Sampled-EfficientZero roots handle");
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    if (r->d_given) { (void)hipFree(r->d_given); r->d_given = nullptr; }
    r->given_records = 0;
    if (r->graph_exec) { (void)hipGraphExecDestroy(r->graph_exec); r->graph_exec = nullptr; }  // the pointers are baked into the graph
    if (!h_draws || records <= 0) return LZ_OK;
    const size_t n = (size_t)records * r->t.B * r->t.A * r->t.D;
    LZ_HIP_CHECK(lz_dev_malloc((void **)&r->d_given, n * 4));
    LZ_HIP_CHECK(hipMemcpyAsync(r->d_given, h_draws, n * 4, hipMemcpyHostToDevice, r->eng->stream));
    LZ_HIP_CHECK(hipStreamSynchronize(r->eng->stream));
    r->given_records = records;
    return LZ_OK;
}

static void launch_dirichlet(lz_roots *r, float alpha, hipStream_t s);

static int prepare_from_inference(lz_roots *r, float root_noise_weight, const float *h_noises_flat, const int32_t *h_to_play,
                                  float device_alpha)
{
    LZ_REQUIRE(r != nullptr && r->inferred && h_to_play != nullptr, "lz_initial_inference must run first; to_play required");
    r->inference_fresh = false;
    const lz_tree_dev &t = r->t;
    if (t.variant == LZ_TREE_SAMPLED_EFFICIENTZERO) {
        // Roots.prepare of the sampled tree (cnode.cpp:640-700): K actions per root drawn from the root (mu | sigma); the
        // Dirichlet noise only perturbs priors that the shipped uniform-prior score never reads
        const size_t B = t.B;
        LZ_HIP_CHECK(hipSetDevice(r->eng->device));
        hipStream_t s = r->eng->stream;
        int largest = h_to_play[0];
        for (size_t i = 1; i < B; ++i) if (h_to_play[i] > largest) largest = h_to_play[i];
        LZ_HIP_CHECK(hipMemcpyAsync(r->d_to_play, h_to_play, B * 4, hipMemcpyHostToDevice, s));
        LZ_HIP_CHECK(hipStreamSynchronize(s));
        lz_sample_args sa;
        sa.given = (r->d_given && r->given_records > 0) ? r->d_given : nullptr;
        sa.policy = r->sim_logits;
        sa.seed = r->seed;
        sa.counter = 0;
        lz_stree_launch_prepare(t, sa, r->d_zero_vp, r->d_to_play, s);
        lz_tree_launch_bump_epoch(t, s);
        LZ_HIP_CHECK(hipGetLastError());
        r->players = largest == -1 ? 1 : 2;
        r->prepared = true;
        r->traverse_count = 0;
        return LZ_OK;
    }
    const size_t B = t.B, A = t.A;
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    hipStream_t s = r->eng->stream;
    int players = 1;
    {
        int largest = h_to_play[0];
        for (size_t i = 1; i < B; ++i) if (h_to_play[i] > largest) largest = h_to_play[i];
        players = largest == -1 ? 1 : 2;
    }
    // everything the kernel needs from the host goes through one pinned buffer and one asynchronous copy: no host-device
    // synchronisation between lz_initial_inference and lz_search (the next graph launch queues behind the network kernels)
    const float *d_noise = nullptr;
    {
        size_t n_noise = 0;
        if (h_noises_flat) {
            if (r->h_n_legal.size() != B) { lz_set_error("legal-action counts of these roots are unknown"); return LZ_ERR_STATE; }
            for (size_t i = 0; i < B; ++i) n_noise += (size_t)r->h_n_legal[i];
            if (n_noise > B * A) { lz_set_error("noise count exceeds root_num * action_space_size"); return LZ_ERR_INVALID; }
        }
        const size_t need = (2 * B + n_noise) * 4;
        if (need > r->prep_bytes) {
            if (r->prep_done) LZ_HIP_CHECK(hipEventSynchronize(r->prep_done));
            if (r->h_prep) (void)hipHostFree(r->h_prep);
            r->h_prep = nullptr; r->prep_bytes = 0;
            LZ_HIP_CHECK(hipHostMalloc(&r->h_prep, need + 4096, hipHostMallocDefault));
            r->prep_bytes = need + 4096;
        }
        if (!r->prep_done) LZ_HIP_CHECK(hipEventCreateWithFlags(&r->prep_done, hipEventDisableTiming));
        else LZ_HIP_CHECK(hipEventSynchronize(r->prep_done));  // the previous upload has left the buffer
        int32_t *hp = (int32_t *)r->h_prep;   // same layout as the device block at d_to_play: to_play | noise offsets | noise
        memcpy(hp, h_to_play, B * 4);
        size_t up = B * 4;
        if (h_noises_flat) {
            int32_t *ho = hp + B;
            size_t acc = 0;
            for (size_t i = 0; i < B; ++i) { ho[i] = (int32_t)acc; acc += (size_t)r->h_n_legal[i]; }
            memcpy(hp + 2 * B, h_noises_flat, n_noise * 4);
            up = (2 * B + n_noise) * 4;
            d_noise = r->d_noise;
        }
        LZ_HIP_CHECK(hipMemcpyAsync(r->d_to_play, hp, up, hipMemcpyHostToDevice, s));
        LZ_HIP_CHECK(hipEventRecord(r->prep_done, s));
    }
    int ragged = 1;
    if (device_alpha > 0.0f) {   // Dirichlet(alpha) over every root's legal actions, drawn on the device: [B][A] by legal position
        launch_dirichlet(r, device_alpha, s);
        d_noise = r->d_noise;
        ragged = 0;
    }
    if (t.variant == LZ_TREE_GUMBEL_MUZERO)  // roots.prepare(noise_w, noises, reward_roots = 0, pred_values, policy_logits, to_play), gumbel_muzero.py:562
        lz_gtree_launch_prepare(t, root_noise_weight, d_noise, ragged, r->d_noise_off, r->d_zero_vp, r->sim_value, r->sim_logits, r->d_to_play, s);
    else
        lz_tree_launch_prepare(t, root_noise_weight, d_noise, ragged, r->d_noise_off, r->d_zero_vp, r->sim_logits, r->d_to_play, s);
    LZ_HIP_CHECK(hipGetLastError());
    r->players = players;
    r->prepared = true;
    r->traverse_count = 0;
    return LZ_OK;
}

extern "C" int lz_roots_prepare_from_inference(lz_roots *r, float root_noise_weight, const float *h_noises_flat,
                                               const int32_t *h_to_play)
{
    return prepare_from_inference(r, root_noise_weight, h_noises_flat, h_to_play, 0.0f);
}

// ---- Dirichlet exploration noise drawn on the device (efficientzero.py:599-602: np.random.dirichlet([alpha] * n_legal) per env).
// One workgroup per root, thread i = position i of the root's legal list: gamma(alpha) by Marsaglia & Tsang (2000) -- for
// alpha < 1 a gamma(alpha + 1) variate times u^(1 / alpha) --, normalised over the root's legal actions.  Counter-based generator
// keyed by (seed, device-resident epoch bumped by every prepare, root, position): a captured or replayed step draws fresh noise
// without any host involvement.  The draws are compared with nothing bit for bit (the reference's come from numpy's
// MT19937); tests/test_dirichlet_gpu.py checks the distribution (Beta marginals, unit sums, independence across roots / steps).
namespace {
// splitmix64 finaliser (same generator as the tree kernels' tie-break / sampling streams, lz_tree_dev.h)
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float u01(uint64_t st, int shift) { return ((float)((st >> shift) & 0xffffff) + 0.5f) * (1.0f / 16777216.0f); }

__global__ __launch_bounds__(256) void k_dirichlet(lz_tree_dev t, float alpha, uint64_t seed, float *__restrict__ out /* [B][A] by legal position */)
{
    __shared__ float s_part[4];
    const int b = blockIdx.x, i = threadIdx.x, n = t.n_legal[b];
    const uint32_t epoch = t.rng_epoch ? t.rng_epoch[0] : 0u;
    float g = 0.0f;
    if (i < n) {
        uint64_t st = mix64(mix64(seed ^ 0xd1b54a32d192ed03ull ^ ((uint64_t)epoch << 20)) ^ ((uint64_t)b << 24) ^ (uint64_t)i);
        const float a = alpha < 1.0f ? alpha + 1.0f : alpha;
        const float d = a - (1.0f / 3.0f), c = __builtin_amdgcn_rsqf(9.0f * d);
        float v = 1.0f;
        for (int it = 0; it < 24; ++it) {   // acceptance > 95 % per round
            st = mix64(st);
            const float u1 = u01(st, 40), u2 = u01(st, 16);
            const float x = __fsqrt_rn(-2.0f * __logf(u1)) * __cosf(6.28318530718f * u2);
            st = mix64(st);
            const float u = u01(st, 40);
            const float w = 1.0f + c * x;
            v = w * w * w;
            if (w > 0.0f && __logf(u) < 0.5f * x * x + d - d * v + d * __logf(v)) break;
            v = 1.0f;
        }
        g = d * v;
        if (alpha < 1.0f) {
            st = mix64(st);
            g *= __expf(__logf(u01(st, 40)) / alpha);
        }
        g = fmaxf(g, 1e-37f);   // (u^(1/alpha) underflows for tiny alpha: keep the sum positive)
    }
    float sum = g;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
    if ((i & 63) == 0) s_part[i >> 6] = sum;
    __syncthreads();
    const float tot = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    if (i < t.A) out[(size_t)b * t.A + i] = i < n ? g / tot : 0.0f;
}
}  // namespace

static void launch_dirichlet(lz_roots *r, float alpha, hipStream_t s)
{
    hipLaunchKernelGGL(k_dirichlet, dim3(r->t.B), dim3(256), 0, s, r->t, alpha, r->seed, r->d_noise);
}

// Roots.prepare with the policy logits of lz_initial_inference AND Dirichlet(alpha) noise drawn on the device for every root's legal
// actions (efficientzero.py:599-605): nothing but to_play crosses PCIe.
extern "C" int lz_roots_prepare_from_inference_dirichlet(lz_roots *r, float root_noise_weight, float root_dirichlet_alpha,
                                                         const int32_t *h_to_play)
{
    LZ_REQUIRE(r != nullptr && r->inferred && h_to_play != nullptr, "lz_initial_inference must run first; to_play required");
    LZ_REQUIRE(root_dirichlet_alpha > 0.0f, "root_dirichlet_alpha must be positive");
    LZ_REQUIRE(r->t.variant != LZ_TREE_SAMPLED_EFFICIENTZERO, "the sampled tree's shipped score never reads the noisy priors: use lz_roots_prepare_from_inference");
    LZ_REQUIRE(r->t.A <= 256, "action space beyond 256");
    return prepare_from_inference(r, root_noise_weight, nullptr, h_to_play, root_dirichlet_alpha);
}

// Timing experiments that SKIP WORK (results are then meaningless) exist only in the -DLZ_DEBUG_KNOBS build
// (liblz_mi355_dbg.so, `python -m lightzero_amd.build --debug-knobs`, used by tools/): LZ_DEBUG_SKIP=<letters> drops launches
// from the search (t tree step, c chain, l LSTM, h heads), LZ_DEBUG_CHAIN_LAYERS cuts the chain short, LZ_DEBUG_CHAIN_TS
// stamps its phases.  The release library contains none of them (tests/test_abi_cpu.py greps the binary).
#ifdef LZ_DEBUG_KNOBS
static unsigned long long *g_chain_ts = nullptr;
extern "C" int lz_debug_read_chain_ts(unsigned long long *h_out)
{
    LZ_REQUIRE(g_chain_ts != nullptr && h_out != nullptr, "LZ_DEBUG_CHAIN_TS was not set");
    LZ_HIP_CHECK(hipDeviceSynchronize());
    LZ_HIP_CHECK(hipMemcpy(h_out, g_chain_ts, 64 * 8, hipMemcpyDeviceToHost));
    return LZ_OK;
}
extern "C" int lz_debug_read_heads_ts(unsigned long long *h_out)   // LZ_DEBUG_HEADS_TS=1: stamps of the last k_heads launch (workgroup 0)
{
    LZ_REQUIRE(lz_debug_heads_ts != nullptr && h_out != nullptr, "LZ_DEBUG_HEADS_TS was not set");
    LZ_HIP_CHECK(hipDeviceSynchronize());
    LZ_HIP_CHECK(hipMemcpy(h_out, lz_debug_heads_ts, 8 * 8, hipMemcpyDeviceToHost));
    return LZ_OK;
}
static unsigned long long *g_tree_ts = nullptr;
extern "C" int lz_debug_read_tree_ts(unsigned long long *h_out)   // LZ_DEBUG_TREE_TS=1: stamps of the tree step in the last fused chain launch
{
    LZ_REQUIRE(g_tree_ts != nullptr && h_out != nullptr, "LZ_DEBUG_TREE_TS was not set");
    LZ_HIP_CHECK(hipDeviceSynchronize());
    LZ_HIP_CHECK(hipMemcpy(h_out, g_tree_ts, 32 * 8, hipMemcpyDeviceToHost));   // [0..6] the tree wave, [8..14] head wave 1 (split heads), [16..31] k_chain_b: layer ends, kernel end
    return LZ_OK;
}
static unsigned long long *g_tree_sep_ts = nullptr;
extern "C" int lz_debug_read_tree_sep_ts(unsigned long long *h_out)   // LZ_DEBUG_TREE_SEP_TS=1: [64 roots][8] stamps of the last separate tree step
{
    LZ_REQUIRE(g_tree_sep_ts != nullptr && h_out != nullptr, "LZ_DEBUG_TREE_SEP_TS was not set");
    LZ_HIP_CHECK(hipDeviceSynchronize());
    LZ_HIP_CHECK(hipMemcpy(h_out, g_tree_sep_ts, 64 * 8 * 8, hipMemcpyDeviceToHost));
    return LZ_OK;
}
static bool dbg_skip(char k)
{
    const char *v = getenv("LZ_DEBUG_SKIP");
    return v && strchr(v, k);
}
#else
static constexpr bool dbg_skip(char) { return false; }
#endif

__global__ void k_stamp_init(unsigned long long *st, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) st[i] = (i & 1) ? 0ull : ~0ull;
}

// the network part of one simulation (mcts_ctree.py:834-847): recurrent_inference for the leaves selected by the
// last traverse, outputs into slot sim + 1 of the pools
// `step` (conv models only): the tree step that selects this simulation's leaves, run inside the chain launch
// the arguments of the chain launch of simulation `sim` (outputs into pool slot sim + 1)
static void chain_args_for(lz_roots *r, int sim, lz_chain_args &ca)
{
    lz_model *m = r->eng->model;
    const lz_model_cfg &c = m->cfg;
    const lz_tree_dev &t = r->t;
    const size_t B = t.B, C = c.num_channels, HW = m->HWl;
    const size_t lat_slot = B * HW * C;
    float *next_latent = r->latent_pool + (size_t)(sim + 1) * lat_slot;
    ca = lz_chain_args{};
    ca.in = r->latent_pool; ca.gather_ix = t.res_ix; ca.slot_stride = (int64_t)lat_slot;
    ca.act_table = m->act_table; ca.action = t.res_last_action; ca.B = (int)B; ca.gw = m->GW; ca.gh = m->GH; ca.C = (int)C;
    ca.layer[ca.nlayers++] = chlayer(m->dyn, 0, 1, 0, act_dyn(c), 1, nullptr);
    const int x_lat = chain_blocks(ca, m->dyn_res, 1, -1, next_latent, act_dyn(c));   // the next latent state: kept for the reward 1x1 conv
    const int x_p = chain_blocks(ca, m->pred_res, x_lat, x_lat, nullptr, act_pred(c));
    ca.c1[0] = c1job(m->val_c, nullptr, r->t_pv, 2 * c.head_channels, 0); ca.c1_in[0] = x_p;
    ca.c1[1] = c1job(m->pol_c, nullptr, r->t_pv, 2 * c.head_channels, c.head_channels); ca.c1_in[1] = x_p;
    ca.c1[2] = c1job(m->rew_c, nullptr, r->t_rx, c.head_channels, 0); ca.c1_in[2] = x_lat;
    ca.nc1 = 3;
    if (m->wide_heads) {   // dense-layer heads read two contiguous [B][HW * HC] blocks
        ca.c1[0] = c1job(m->val_c, nullptr, r->t_pv, c.head_channels, 0);
        ca.c1[1] = c1job(m->pol_c, nullptr, r->t_pv + B * HW * c.head_channels, c.head_channels, 0);
    }
    ca.c1[0].act = ca.c1[1].act = act_pred(c);
    ca.c1[2].act = act_dyn(c);
    ca.gelu = act_pred(c) == 2 || act_dyn(c) == 2;
}

// split heads: what the chain launch that follows simulation `leaf_slot - 1` needs to finish that simulation's heads (lz_split_heads)
static void split_heads_for(lz_roots *r, int leaf_slot, lz_split_heads &sh)
{
    lz_model *m = r->eng->model;
    const size_t B = r->t.B, A = r->t.A;
    sh = lz_split_heads{};
    sh.on = 1;
    sh.part = r->sh_part;
    const MlpW *w[3] = {&m->fc_value, &m->fc_policy, &m->fc_reward};
    for (int i = 0; i < 3; ++i) { sh.b1[i] = w[i]->b1; sh.s1[i] = w[i]->s1; sh.t1[i] = w[i]->t1; sh.w2t[i] = w[i]->w2; sh.b2[i] = w[i]->b2; }
    sh.nout = m->cfg.support_size; sh.n_unit_tiles = m->cfg.lstm_hidden_size / 16; sh.support_min = m->cfg.support_min;
    sh.out_value = r->sim_value + (size_t)leaf_slot * B; sh.out_vp = r->sim_vp + (size_t)leaf_slot * B;
    sh.out_logits = r->sim_logits + (size_t)leaf_slot * B * A;
    if (r->trace_on && r->head_debug && r->hd_logits && r->hd_sup == (size_t)sh.nout) {
        // [2][B][nout] / [2][B] of the leaf's slot; the kernel indexes (head, root) itself
        sh.dbg_logits = hd_logits_at(r, leaf_slot, 0); sh.dbg_expect = hd_expect_at(r, leaf_slot, 0); sh.dbg_B = (int)B;
    }
}

// One launch per simulation (LZ_SIM_ONE_LAUNCH=1): `hold` != null -- the LSTM launch of this simulation is not enqueued, its arguments are
// returned; `held` != null -- the LSTM launch of the PREVIOUS simulation rides in front of this simulation's chain launch (k_sim_fused,
// launch number `fuse_launch` of this search), or is enqueued by itself first where the fused form does not apply.
static void recurrent(lz_roots *r, int sim, int horizon, hipStream_t s, const lz_tree_step *step = nullptr, bool defer_heads = false,
                      const lz_lstm_args *held = nullptr, lz_lstm_args *hold = nullptr, int *fuse_launch = nullptr)
{
    lz_model *m = r->eng->model;
    if (m->cfg.model_type >= 2) { lz_mlp_recurrent(r, sim, horizon, s); return; }
    const lz_model_cfg &c = m->cfg;
    const lz_tree_dev &t = r->t;
    const size_t B = t.B, A = policy_width_of(m, t), HW = m->HWl, H = c.lstm_hidden_size;
    const int slot = sim + 1;
    // ---- dynamics conv over [latent | one-hot action] + BN + latent + ReLU, dynamics residual block (-> latent pool
    // slot), prediction residual block and the three 1x1 head convs (efficientzero_model.py:527-558, common.py:1189-1203):
    // ONE launch, activations stay in LDS
    {
        lz_chain_args ca;
        chain_args_for(r, sim, ca);
#ifdef LZ_DEBUG_KNOBS
        if (const char *dbg = getenv("LZ_DEBUG_CHAIN_LAYERS")) ca.nlayers = atoi(dbg);  // timing experiments only
        if (const char *dbg = getenv("LZ_DEBUG_CHAIN_FLAGS")) ca.debug_flags = atoi(dbg);
        if (getenv("LZ_DEBUG_HEADS_TS") && !lz_debug_heads_ts) (void)lz_dev_malloc((void **)&lz_debug_heads_ts, 8 * 8);
        if (getenv("LZ_DEBUG_CHAIN_TS")) {  // timing experiments only: stamps of the last launch, read with lz_debug_read_chain_ts
            if (!g_chain_ts) (void)lz_dev_malloc((void **)&g_chain_ts, 64 * 8);
            ca.tstamp = g_chain_ts;
        }
#endif
        if (step && !lz_chain_fusable(ca, *step)) {  // the tree outgrew the LDS budget (or the shape has no fused instance)
            // (never with split heads: the simulation before this one deferred its heads only after the same test succeeded)
            lz_tree_launch_backprop_traverse(step->t, step->new_node, step->discount, step->vps, step->values, step->logits,
                                             step->horizon, step->a, step->delta, step->vtp, s);
            step = nullptr;
        }
        // trace: the selection this simulation's network launches consume.  With the tree step fused into the chain launch the
        // res_* arrays are written by that launch's prologue, so the copy follows it (same stream; also inside a captured graph)
        if (r->trace_on && !step) (void)hipMemcpyAsync(r->trace + (size_t)sim * 5 * B, t.res_ix, 5 * B * 4, hipMemcpyDeviceToDevice, s);
        if (r->stamps_on && r->stamps) ca.stamp = r->stamps + (size_t)slot * 4;
        {
            ProfScope ps(r->eng, s);
            bool fused = false;
            if (held) {
                fused = step && !r->trace_on && !ca.stamp && !ca.tstamp && fuse_launch && r->fuse_ctl &&
                        lz_launch_sim_fused(*held, ca, *step, r->fuse_ctl, *fuse_launch, s);
                if (fused) { ++*fuse_launch; r->fuse_used = true; }
                else lz_launch_lstm(*held, s);
            }
            if (!fused && !dbg_skip('c')) lz_launch_chain(ca, s, step);
        }
        if (r->trace_on && step) (void)hipMemcpyAsync(r->trace + (size_t)sim * 5 * B, t.res_ix, 5 * B * 4, hipMemcpyDeviceToDevice, s);
    }
    // ---- value prefix LSTM (+ BN1d + ReLU), then the three head MLPs with h^-1 fused
    lz_lstm_args l{};
    l.x = r->t_rx; l.h_pool = r->h_pool; l.c_pool = r->c_pool; l.gather_ix = t.res_ix; l.wcat = m->lstm_w; l.wf = m->lstm_wf; l.wb = m->lstm_wb; l.bias = m->lstm_b;
    l.bn_scale = m->vp_s; l.bn_shift = m->vp_t; l.search_len = t.res_search_len; l.horizon = horizon;
    l.h_out = r->h_pool + (size_t)slot * B * H; l.c_out = r->c_pool + (size_t)slot * B * H; l.hbn_out = r->t_hbn;
    l.B = (int)B; l.KX = c.head_channels * (int)HW; l.H = (int)H;
    l.gelu = act_dyn(c) == 2;
    if (defer_heads) {   // split heads: the first layers of the head MLPs ride on this launch, the next chain launch finishes them
        l.sh_pv = r->t_pv; l.sh_kc = 2 * c.head_channels * (int)HW; l.sh_w1c = m->sh_w1c; l.sh_w1r = m->sh_w1r;
        l.sh_part = r->sh_part;
    }
#ifdef LZ_DEBUG_KNOBS
    l.debug_hot_weights = getenv("LZ_DEBUG_LSTM_HOTW") ? atoi(getenv("LZ_DEBUG_LSTM_HOTW")) : 0;   // bit 0: hot weights; split heads: bit 1 no operand loads, bit 2 no partial stores
#endif
    // (running the value / policy heads on a side stream beside the LSTM was measured: the cross-stream
    // dependencies cost more than the overlap gains, 6.7 vs 5.8 ms per step)
    if (r->stamps_on && r->stamps) l.stamp = r->stamps + (size_t)slot * 4 + 2;
    if (hold && defer_heads && c.model_type == 0) *hold = l;     // rides in front of the next chain launch
    else if (c.model_type == 0 && !dbg_skip('l')) lz_launch_lstm(l, s);
    if (!defer_heads && !dbg_skip('h'))
        heads(r, slot, r->sim_value + (size_t)slot * B, r->sim_logits + (size_t)slot * B * A, r->dbg_logits[0], true,
              r->sim_vp + (size_t)slot * B, r->dbg_logits[1], s);
}

// the exploration-factor table of lz_traverse_args::tab: a function of (pb_c_base, pb_c_init) -- computed when they change, not once per search
// (it was a 4.6 us launch at the head of every captured search)
static void ensure_explore_tab(lz_roots *r, int pb_c_base, float pb_c_init, hipStream_t s)
{
    if (!r->explore_tab || (r->tab_valid && r->tab_base == pb_c_base && r->tab_init == pb_c_init)) return;
    lz_tree_launch_explore_tab(r->explore_tab, pb_c_base, pb_c_init, s);
    r->tab_base = pb_c_base; r->tab_init = pb_c_init; r->tab_valid = true;
}

// the whole search: traverse, then per simulation [network, expand + backup fused with the next selection]
static void enqueue_search(lz_roots *r, int num_simulations, lz_traverse_args ta, float delta, int horizon, hipStream_t s)
{
    const lz_tree_dev &t = r->t;
    const size_t B = t.B;
    const size_t A = policy_width_of(r->eng->model, t);
    ta.counter = 0;
    if (r->explore_tab) ta.tab = r->explore_tab;   // filled by lz_search (ensure_explore_tab) OUTSIDE the captured sequence: it depends on the search parameters only
    if (t.variant == LZ_TREE_SAMPLED_EFFICIENTZERO) {
        lz_tree_launch_minmax_reset(t, s);  // a fresh MinMaxStatsList per search (mcts_ctree.py:778-779)
        // SampledEfficientZeroMCTSCtree.search (mcts_ctree_sampled.py:480-600): the leaf's K actions are drawn on the
        // device from the (mu | sigma) the network just produced (or copied from the injected draws of a parity run)
        const size_t KD = (size_t)t.A * t.D;
        ta.counter = 0;
        lz_stree_launch_traverse(t, ta, delta, r->d_to_play, s);
        for (int sim = 0; sim < num_simulations; ++sim) {
            recurrent(r, sim, horizon, s);
            const int slot = sim + 1;
            lz_sample_args sa;
            sa.given = (r->d_given && slot < r->given_records) ? r->d_given + (size_t)slot * B * KD : nullptr;
            sa.policy = r->sim_logits + (size_t)slot * B * A;
            sa.seed = r->seed;
            sa.counter = (uint32_t)slot;
            const float *vp = r->sim_vp + (size_t)slot * B, *val = r->sim_value + (size_t)slot * B;
            if (sim + 1 < num_simulations) {
                ta.counter = (uint32_t)(sim + 1);
                lz_stree_launch_backprop_traverse(t, slot, ta.discount, vp, val, sa, horizon, ta, delta, r->d_to_play, s);
            } else {
                lz_stree_launch_backprop(t, slot, ta.discount, vp, val, sa, nullptr, horizon, nullptr, s);
            }
        }
        return;
    }
    if (r->stamps_on && r->stamps)   // {start, end} x {chain, LSTM} per pool slot, cleared per search
        hipLaunchKernelGGL(k_stamp_init, dim3((unsigned)((t.NN * 4 + 255) / 256)), dim3(256), 0, s, r->stamps, t.NN * 4);
#ifdef LZ_DEBUG_KNOBS
    if (getenv("LZ_DEBUG_TREE_SEP_TS")) {
        if (!g_tree_sep_ts) (void)lz_dev_malloc((void **)&g_tree_sep_ts, 64 * 8 * 8);
        ta.dbg_ts = g_tree_sep_ts;
    }
#endif
    ta.fresh_minmax = 1;   // ... which the first selection starts itself
    lz_tree_launch_traverse(t, ta, delta, r->d_to_play, s);
    ta.fresh_minmax = 0;
    // The expand + backup of simulation s and the selection of simulation s + 1 are one tree step per root; for the conv
    // models it runs in the prologue of simulation s + 1's chain launch (same workgroup-per-root mapping) while the tree
    // of a root still fits the LDS budget, else as its own launch.  LZ_NO_TREE_FUSE=1 keeps it separate (parity tests).
    const bool fuse = r->eng->model->cfg.model_type < 2 && !getenv("LZ_NO_TREE_FUSE");
    // Split heads (EfficientZero, 6x6 latent; LZ_HEADS_LAUNCH=1 keeps the separate head launch): a simulation whose successor's chain
    // launch will run the tree step in its prologue leaves its heads to that launch -- the LSTM launch computes their first layers as
    // partial sums, waves 1..7 of the next chain launch finish them for their root while wave 0 stages the tree.  2 launches per
    // simulation instead of 3; the last simulation (and every one followed by a separate tree launch) keeps the head launch.
    lz_model *mdl = r->eng->model;
    const bool split = fuse && mdl->cfg.model_type == 0 && mdl->sh_w1c && r->sh_part && t.A <= 64 && mdl->cfg.support_size <= 768 &&
                       mdl->cfg.lstm_hidden_size == 512 && t.variant == LZ_TREE_EFFICIENTZERO && !getenv("LZ_HEADS_LAUNCH") && !getenv("LZ_CHAIN_DIRECT") &&
                       !getenv("LZ_CHAIN_W4") && !getenv("LZ_LSTM_NOSPLIT") && !getenv("LZ_LSTM_ROWS32") && !getenv("LZ_LSTM_CHUNKED") &&
                       !(mdl->GW == 8 && getenv("LZ_CHAIN_NO_SPLIT"))   // (8x8: only k_chain_s3g has the split-head instance)
#ifdef LZ_DEBUG_KNOBS
                       // recurrent() mutates the chain arguments under these switches AFTER the defer decision below was taken from the
                       // unmodified ones (a stamped / truncated chain is not fusable): a simulation could defer its heads to a launch
                       // that then never finishes them.  The timing experiments that use them run with the separate head launch.
                       && !getenv("LZ_DEBUG_CHAIN_TS") && !getenv("LZ_DEBUG_CHAIN_LAYERS") && !getenv("LZ_DEBUG_CHAIN_FLAGS")
#endif
        ;
    auto make_step = [&](int slot) {
        lz_tree_step st{};
        st.t = t; st.new_node = slot; st.discount = ta.discount;
        st.vps = r->sim_vp + (size_t)slot * B; st.values = r->sim_value + (size_t)slot * B; st.logits = r->sim_logits + (size_t)slot * B * A;
        st.horizon = horizon; st.a = ta; st.delta = delta; st.vtp = r->d_to_play;
        return st;
    };
    // One launch per simulation (opt-in; k_sim_fused in lz_nn.hip): where a simulation defers its heads, its LSTM launch is held back and becomes
    // the first phase of the next simulation's chain launch.  Not with tracing or stamps (their copies / stamp words sit between the launches).
    static const char *one_launch_env = getenv("LZ_SIM_ONE_LAUNCH");
    const bool one_launch = one_launch_env && atoi(one_launch_env) != 0 && split && r->fuse_ctl && !r->trace_on && !(r->stamps_on && r->stamps) && (B == 256 || B == 128) &&
                            lz_fused_ctl_bytes((int)B, num_simulations) <= r->fuse_ctl_bytes && !getenv("LZ_CHAIN_NO_SPLIT");
    r->fuse_used = false;
    if (one_launch) {   // (a kernel, not hipMemsetAsync: as a memset NODE of the captured graph the clear was observed not to happen on a later replay)
        const int nw = (int)(lz_fused_ctl_bytes((int)B, num_simulations) / 4);
        hipLaunchKernelGGL(k_zero_words, dim3((nw + 255) / 256), dim3(256), 0, s, reinterpret_cast<unsigned *>(r->fuse_ctl), nw);
    }
    lz_lstm_args held_lstm{}, next_lstm{};
    bool have_held = false;
    int fuse_launch = 0;
    lz_tree_step step{};
    bool pending = false;  // a step that the next chain launch has to run
    for (int sim = 0; sim < num_simulations; ++sim) {
        bool defer = false;   // will simulation sim + 1's chain launch run the tree step -- and with it this simulation's heads?
        if (split && sim + 1 < num_simulations) {
            lz_traverse_args ta2 = ta;
            ta2.counter = (uint32_t)(sim + 1);
            lz_tree_step nxt = make_step(sim + 1);
            nxt.a = ta2;
            lz_chain_args ca;
            chain_args_for(r, sim + 1, ca);
            defer = lz_chain_fusable(ca, nxt);
        }
        const bool hold_now = one_launch && defer;
        recurrent(r, sim, horizon, s, pending ? &step : nullptr, defer, have_held ? &held_lstm : nullptr, hold_now ? &next_lstm : nullptr, &fuse_launch);
        have_held = hold_now;
        if (hold_now) held_lstm = next_lstm;
        pending = false;
        const int slot = sim + 1;
        const float *vp = r->sim_vp + (size_t)slot * B, *val = r->sim_value + (size_t)slot * B, *lg = r->sim_logits + (size_t)slot * B * A;
        // is_reset = search_len % horizon == 0 is derived on the device (mcts_ctree.py:859)
        if (sim + 1 < num_simulations) {
            ta.counter = (uint32_t)(sim + 1);
            if (fuse) {
                step = make_step(slot);
                if (defer) split_heads_for(r, slot, step.sh);
#ifdef LZ_DEBUG_KNOBS
                if (getenv("LZ_DEBUG_TREE_TS")) {   // timing experiments only: stamps of root 0's step in the last fused launch
                    if (!g_tree_ts) (void)lz_dev_malloc((void **)&g_tree_ts, 32 * 8);
                    step.ts = g_tree_ts;
                }
#endif
                pending = true;
            } else if (!dbg_skip('t')) {
                lz_tree_launch_backprop_traverse(t, slot, ta.discount, vp, val, lg, horizon, ta, delta, r->d_to_play, s);
            }
        } else {
            lz_tree_launch_backprop(t, slot, ta.discount, vp, val, lg, nullptr, horizon, nullptr, s);
        }
    }
}

// the debugging switches that change a captured launch sequence (so that toggling one re-captures)
static uint64_t graph_knobs()
{
    uint64_t knobs = 0;
    const char *names[] = {"LZ_TRAVERSE_SERIAL", "LZ_TREE_NO_WG", "LZ_NO_TREE_FUSE", "LZ_TREE_NO_LDS", "LZ_TREE_LDS_LIMIT", "LZ_LSTM_CHUNKED", "LZ_HEADS_256",
#ifdef LZ_DEBUG_KNOBS
                           "LZ_DEBUG_SKIP",
#endif
                           "LZ_LSTM_ROWS32", "LZ_LSTM_NOSPLIT", "LZ_LSTM3", "LZ_CONV_DIRECT", "LZ_CONV_NO_SPLIT", "LZ_CONV_NO_DUAL", "LZ_CHAIN_NO_SPLIT", "LZ_CHAIN_DIRECT", "LZ_CHAIN_W4", "LZ_HEADS_VALU", "LZ_HEADS_LAUNCH", "LZ_SIM_ONE_LAUNCH", "LZ_LSTM_NO_OVL", "LZ_HEADS_MM64", "LZ_TREE_WIDE"};
    for (const char *n : names) {
        const char *v = getenv(n);
        knobs = knobs * 1000003ull + 7;
        for (; v && *v; ++v) knobs = knobs * 131ull + (unsigned char)*v;
    }
    return knobs;
}

// The launch sequence of a fused search depends only on its parameters (every pointer is a fixed offset into the roots'
// slabs), so it is captured once into a HIP graph per (roots, key) and replayed: the launches of all simulations become one
// graph launch, and the host leaves the loop.
template <class F>
static int launch_captured(lz_roots *r, const lz_graph_key &key, F enqueue)
{
    hipStream_t s = r->eng->stream;
    if (!r->graph_exec || memcmp(&key, &r->graph_key, sizeof(key)) != 0) {
        if (r->graph_exec) { (void)hipGraphExecDestroy(r->graph_exec); r->graph_exec = nullptr; }
        hipGraph_t g = nullptr;
        LZ_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        enqueue();
        hipError_t e1 = hipStreamEndCapture(s, &g);
        if (e1 != hipSuccess || !g) { lz_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e1)); return LZ_ERR_HIP; }
        hipError_t e2 = hipGraphInstantiate(&r->graph_exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e2 != hipSuccess) { r->graph_exec = nullptr; lz_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e2)); return LZ_ERR_HIP; }
        r->graph_key = key;
    }
    LZ_HIP_CHECK(hipGraphLaunch(r->graph_exec, s));
    return LZ_OK;
}

extern "C" int lz_search(lz_roots *r, int num_simulations, int pb_c_base, float pb_c_init, float discount_factor,
                         int lstm_horizon_len, float value_delta_max)
{
    LZ_REQUIRE(r != nullptr, "roots is NULL");
    LZ_REQUIRE(r->inferred && r->prepared, "lz_search needs lz_initial_inference and a prepare call first");
    LZ_REQUIRE(r->pool_model_uid == r->eng->model_uid, "the engine's model was replaced after these roots were inferred");
    LZ_REQUIRE(lstm_horizon_len > 0 || r->eng->model->cfg.model_type == 1 || r->eng->model->cfg.model_type == 2, "lstm_horizon_len must be positive (mcts_ctree.py:858)");
    if (num_simulations < 1 || num_simulations >= r->t.NN) {
        lz_set_error("num_simulations %d exceeds the node pool (max_simulations %d)", num_simulations, r->t.NN - 1);
        return LZ_ERR_STATE;
    }
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    hipStream_t s = r->eng->stream;
    if (r->trace_on && r->head_debug && r->eng->model->cfg.model_type < 2)   // (allocations cannot happen inside the capture)
        if (int rc = ensure_head_debug(r)) return rc;
    if (r->stamps_on && !r->stamps) LZ_HIP_CHECK(lz_dev_malloc((void **)&r->stamps, (size_t)r->t.NN * 4 * sizeof(unsigned long long)));
    r->delta = value_delta_max;
    lz_traverse_args ta;
    ta.pb_c_base = pb_c_base; ta.pb_c_init = pb_c_init; ta.discount = discount_factor; ta.players = r->players;
    ta.tiebreak = r->tiebreak; ta.seed = r->seed; ta.counter = 0;
    ta.serial = getenv("LZ_TRAVERSE_SERIAL") ? 1 : 0;
    const bool use_graph = !r->eng->prof_on && !getenv("LZ_NO_GRAPH");
    // LZ_SIM_ONE_LAUNCH=1: a fused launch's group barrier is a BOUNDED spin; one that ran out (a workgroup not resident, an XCD with another
    // share of the workgroups) raised lz_res_ctl::fault and the search's results are void -- reported here, synchronously (the opt-in mode
    // gives up the host's overlap with the search for this check), so that the caller can repeat the env-step on the two-launch path
    auto check_fused = [&]() -> int {
        if (!r->fuse_used || !r->fuse_ctl) return LZ_OK;
        static const char *mode = getenv("LZ_SIM_ONE_LAUNCH");
        if (mode && atoi(mode) == 2) return LZ_OK;   // timing experiments only: no synchronous fault check (a fault then goes unnoticed)
        unsigned hdr[17] = {0};
        LZ_HIP_CHECK(hipStreamSynchronize(s));
        LZ_HIP_CHECK(hipMemcpy(hdr, r->fuse_ctl, sizeof(hdr), hipMemcpyDeviceToHost));
        if (hdr[16]) {
            lz_set_error("LZ_SIM_ONE_LAUNCH: %u workgroup(s) of a fused launch gave up waiting for their 16-root group (the launch needs one resident "
                         "workgroup per root and the dispatcher's round-robin over the XCDs; tickets per XCD %u %u %u %u %u %u %u %u); the search's "
                         "results are void -- unset LZ_SIM_ONE_LAUNCH and repeat the env-step", hdr[16], hdr[0], hdr[1], hdr[2], hdr[3], hdr[4], hdr[5], hdr[6], hdr[7]);
            return LZ_ERR_STATE;
        }
        return LZ_OK;
    };
    ensure_explore_tab(r, pb_c_base, pb_c_init, s);
    if (!use_graph) {
        enqueue_search(r, num_simulations, ta, value_delta_max, lstm_horizon_len, s);
        LZ_HIP_CHECK(hipGetLastError());
        return check_fused();
    }
    lz_graph_key key{};  // value-initialised: the padding takes part in the memcmp
    key.sims = num_simulations; key.pb_c_base = pb_c_base; key.pb_c_init = pb_c_init; key.discount = discount_factor;
    key.horizon = lstm_horizon_len; key.delta = value_delta_max; key.players = r->players; key.tiebreak = r->tiebreak;
    key.seed = r->seed; key.knobs = graph_knobs();
    key.model_uid = r->eng->model_uid; key.weights_gen = r->eng->weights_gen; key.trace = (r->trace_on ? 1 : 0) | (r->trace_on && r->head_debug ? 2 : 0); key.stamps = r->stamps_on ? 1 : 0;
    if (int rc = launch_captured(r, key, [&]() { enqueue_search(r, num_simulations, ta, value_delta_max, lstm_horizon_len, s); })) return rc;
    return check_fused();
}

// GumbelMuZeroMCTSCtree.search (mcts_ctree.py:1067-1172) with an engine MuZero model: sequential-halving selection, MuZero
// recurrent inference, expand + backup fused with the next selection, all on the device
extern "C" int lz_gsearch(lz_roots *r, int num_simulations, int max_num_considered_actions, float discount_factor)
{
    LZ_REQUIRE(r != nullptr, "roots is NULL");
    LZ_REQUIRE(r->t.variant == LZ_TREE_GUMBEL_MUZERO, "not a Gumbel MuZero roots handle");
    LZ_REQUIRE(r->inferred && r->prepared, "lz_gsearch needs lz_initial_inference and a prepare call first");
    LZ_REQUIRE(r->pool_model_uid == r->eng->model_uid, "the engine's model was replaced after these roots were inferred");
    LZ_REQUIRE(r->players == 1, "the Gumbel MuZero tree is single-player (cnode.cpp:618)");
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    hipStream_t s = r->eng->stream;
    int rc = lz_groots_set_considered(r, num_simulations, max_num_considered_actions, s);
    if (rc != LZ_OK) return rc;
    const lz_tree_dev &t = r->t;
    const size_t B = t.B, A = policy_width_of(r->eng->model, t);
    auto enqueue = [&]() {
        lz_tree_launch_minmax_reset(t, s);
        lz_gtree_launch_traverse(t, discount_factor, s);
        for (int sim = 0; sim < num_simulations; ++sim) {
            recurrent(r, sim, 0, s);
            const int slot = sim + 1;
            const float *rew = r->sim_vp + (size_t)slot * B, *val = r->sim_value + (size_t)slot * B, *lg = r->sim_logits + (size_t)slot * B * A;
            if (sim + 1 < num_simulations) lz_gtree_launch_backprop_traverse(t, slot, discount_factor, rew, val, lg, s);
            else lz_gtree_launch_backprop(t, slot, discount_factor, rew, val, lg, s);
        }
    };
    if (r->eng->prof_on || getenv("LZ_NO_GRAPH")) {
        enqueue();
        LZ_HIP_CHECK(hipGetLastError());
        return LZ_OK;
    }
    lz_graph_key key{};
    key.sims = num_simulations; key.pb_c_base = max_num_considered_actions; key.discount = discount_factor;
    key.horizon = -7;  // marks a Gumbel search (a roots handle is either Gumbel or not, so the slot is never shared)
    key.players = r->players; key.tiebreak = r->tiebreak; key.seed = r->seed; key.knobs = graph_knobs();
    key.model_uid = r->eng->model_uid; key.weights_gen = r->eng->weights_gen; key.trace = (r->trace_on ? 1 : 0) | (r->trace_on && r->head_debug ? 2 : 0); key.stamps = r->stamps_on ? 1 : 0;
    return launch_captured(r, key, enqueue);
}

// ReZero: EfficientZeroMCTSCtree.search_with_reuse / MuZeroMCTSCtree.search_with_reuse (mcts_ctree.py:878-1002, 370-470) with an
// engine model.  Every root goes through the network kernels each simulation (a root that needs no inference only wastes
// its lane of the batch; its outputs land in a pool slot no node refers to), the tree kernels implement the reuse rules.
extern "C" int lz_search_with_reuse(lz_roots *r, int num_simulations, int pb_c_base, float pb_c_init, float discount_factor,
                                    int lstm_horizon_len, float value_delta_max, const int32_t *h_true_action,
                                    const float *h_reuse_value, int *out_last_length, double *out_average_infer)
{
    LZ_REQUIRE(r != nullptr && h_true_action != nullptr && h_reuse_value != nullptr, "NULL argument");
    LZ_REQUIRE(r->inferred && r->prepared, "lz_search_with_reuse needs lz_initial_inference and a prepare call first");
    LZ_REQUIRE(r->pool_model_uid == r->eng->model_uid, "the engine's model was replaced after these roots were inferred");
    LZ_REQUIRE(r->t.variant != LZ_TREE_SAMPLED_EFFICIENTZERO, "the sampled tree has no reuse variant");
    const int mt = r->eng->model->cfg.model_type;
    LZ_REQUIRE(lstm_horizon_len > 0 || mt == 1 || mt == 2, "lstm_horizon_len must be positive (mcts_ctree.py:967)");
    if (num_simulations < 1 || num_simulations >= r->t.NN) {
        lz_set_error("num_simulations %d exceeds the node pool (max_simulations %d)", num_simulations, r->t.NN - 1);
        return LZ_ERR_STATE;
    }
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    hipStream_t s = r->eng->stream;
    const lz_tree_dev &t = r->t;
    const size_t B = t.B, A = policy_width_of(r->eng->model, t);
    if (!r->d_reuse) LZ_HIP_CHECK(lz_dev_malloc((void **)&r->d_reuse, (2 * B + (size_t)t.NN) * 4));
    int32_t *d_ta = (int32_t *)r->d_reuse;
    float *d_rv = (float *)(d_ta + B);
    int32_t *d_cnt = (int32_t *)(d_rv + B);
    LZ_HIP_CHECK(hipMemcpyAsync(d_ta, h_true_action, B * 4, hipMemcpyHostToDevice, s));
    LZ_HIP_CHECK(hipMemcpyAsync(d_rv, h_reuse_value, B * 4, hipMemcpyHostToDevice, s));
    LZ_HIP_CHECK(hipMemsetAsync(d_cnt, 0, (size_t)t.NN * 4, s));
    r->delta = value_delta_max;
    lz_traverse_args ta;
    ta.pb_c_base = pb_c_base; ta.pb_c_init = pb_c_init; ta.discount = discount_factor; ta.players = r->players;
    ta.tiebreak = r->tiebreak; ta.seed = r->seed; ta.counter = 0;
    lz_tree_launch_minmax_reset(t, s);
    for (int sim = 0; sim < num_simulations; ++sim) {
        ta.counter = (uint32_t)sim;
        lz_tree_launch_traverse_reuse(t, ta, value_delta_max, r->d_to_play, d_ta, d_rv, s);
        recurrent(r, sim, lstm_horizon_len, s);
        const int slot = sim + 1;
        lz_tree_launch_backprop_reuse(t, slot, discount_factor, r->sim_vp + (size_t)slot * B, r->sim_value + (size_t)slot * B,
                                      r->sim_logits + (size_t)slot * B * A, nullptr, lstm_horizon_len, nullptr, nullptr, nullptr, d_rv,
                                      d_ta, d_cnt + sim, s);
    }
    LZ_HIP_CHECK(hipGetLastError());
    std::vector<int32_t> cnt(num_simulations);
    LZ_HIP_CHECK(hipMemcpyAsync(cnt.data(), d_cnt, (size_t)num_simulations * 4, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    long long sum = 0;
    for (int v : cnt) sum += v;
    if (out_last_length) *out_last_length = cnt.back();
    if (out_average_infer) *out_average_infer = (double)sum / num_simulations;
    return LZ_OK;
}

// ------------------------------------------------------------------------------------------------
extern "C" int lz_roots_enable_trace(lz_roots *r, int on)
{
    LZ_REQUIRE(r != nullptr, "roots is NULL");
    r->trace_on = on != 0;
    r->head_debug = (on & 2) != 0;   // + per-slot support-wide head logits / pre-transform expectations of EVERY simulation
    return LZ_OK;
}

// In-graph timing of the two launches of a simulation (bench.py's roofline clock): while on, the chain and LSTM
// launches' first / last workgroup store their s_memrealtime start / end into [slot][4] words; the captured search is re-captured with the stamp pointers.
extern "C" int lz_roots_enable_stamps(lz_roots *r, int on)
{
    LZ_REQUIRE(r != nullptr, "roots is NULL");
    r->stamps_on = on != 0;
    return LZ_OK;
}

// h_out [num_simulations][4] = {chain: first workgroup start, last workgroup end, LSTM: start, end} of simulation s (pool slot s + 1),
// in ticks of the 100 MHz constant-rate counter (10 ns)
extern "C" int lz_roots_read_stamps(lz_roots *r, int num_simulations, uint64_t *h_out)
{
    LZ_REQUIRE(r != nullptr && h_out != nullptr && r->stamps != nullptr, "no stamps: lz_roots_enable_stamps(roots, 1) before lz_search");
    LZ_REQUIRE(num_simulations >= 1 && num_simulations < r->t.NN, "num_simulations out of range");
    hipStream_t s = r->eng->stream;
    LZ_HIP_CHECK(hipMemcpyAsync(h_out, r->stamps + 4, (size_t)num_simulations * 4 * 8, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    return LZ_OK;
}

// Head debug buffers (lz_roots_enable_trace(roots, 3) before the inference / search): the support-wide logits [B][support] and the
// pre-transform expectation softmax . support [B] of the value (which = 0) or value-prefix / reward (which = 1) head at pool slot
// `slot` -- for EVERY simulation, whichever kernel finished the head (k_heads_mm, or the split heads in the next chain launch)
extern "C" int lz_roots_read_head_debug(lz_roots *r, int slot, int which, float *h_logits, float *h_expect)
{
    LZ_REQUIRE(r != nullptr && (which == 0 || which == 1), "bad argument");
    LZ_REQUIRE(r->trace_on && r->head_debug && r->hd_logits, "head debug is off: lz_roots_enable_trace(roots, 3) before the inference");
    LZ_REQUIRE(slot >= 0 && slot < r->t.NN, "slot out of range");
    const lz_model_cfg &mc = r->eng->model->cfg;
    const size_t B = r->t.B, SUP = (which == 1 && mc.reward_support_size > 0) ? mc.reward_support_size : mc.support_size;
    hipStream_t s = r->eng->stream;
    if (h_logits) LZ_HIP_CHECK(hipMemcpyAsync(h_logits, hd_logits_at(r, slot, which), B * SUP * 4, hipMemcpyDeviceToHost, s));
    if (h_expect) LZ_HIP_CHECK(hipMemcpyAsync(h_expect, hd_expect_at(r, slot, which), B * 4, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    return LZ_OK;
}

// out[i] = the device's inverse scalar transform of in[i] (lz_hinv.h) -- host arrays in and out.  which = 0: the copy compiled into the
// conv-model head kernels (k_heads, k_heads_mm, split heads), 1: the copy in the MLP family's k_rowfinal.  tests/test_hinv_gpu.py holds
// both bit-equal to torch's evaluation of scaling_transform.py:88-91.
extern "C" int lz_debug_inverse_scalar_transform(lz_engine *e, int which, const float *h_in, int64_t n, float *h_out)
{
    LZ_REQUIRE(e != nullptr && h_in != nullptr && h_out != nullptr && n > 0 && (which == 0 || which == 1), "bad argument");
    LZ_HIP_CHECK(hipSetDevice(e->device));
    float *d = nullptr;
    LZ_HIP_CHECK(lz_dev_malloc((void **)&d, (size_t)n * 8));
    hipStream_t s = e->stream;
    hipError_t err = hipMemcpyAsync(d, h_in, (size_t)n * 4, hipMemcpyHostToDevice, s);
    if (err == hipSuccess) {
        if (which == 0) lz_launch_hinv_nn(d, d + n, n, s); else lz_launch_hinv_dense(d, d + n, n, s);
        err = hipMemcpyAsync(h_out, d + n, (size_t)n * 4, hipMemcpyDeviceToHost, s);
    }
    if (err == hipSuccess) err = hipStreamSynchronize(s);
    (void)hipFree(d);
    LZ_HIP_CHECK(err);
    return LZ_OK;
}

extern "C" int lz_roots_read_trace(lz_roots *r, int num_simulations, int32_t *h_out)
{
    LZ_REQUIRE(r != nullptr && h_out != nullptr && r->trace != nullptr, "no trace");
    LZ_REQUIRE(r->trace_on, "tracing is off: call lz_roots_enable_trace(roots, 1) before lz_search");
    LZ_REQUIRE(num_simulations >= 1 && num_simulations < r->t.NN, "num_simulations out of range");
    const size_t B = r->t.B;
    std::vector<int32_t> tmp((size_t)num_simulations * 5 * B);
    hipStream_t s = r->eng->stream;
    LZ_HIP_CHECK(hipMemcpyAsync(tmp.data(), r->trace, tmp.size() * 4, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    for (int sim = 0; sim < num_simulations; ++sim)
        for (size_t b = 0; b < B; ++b) {
            const int32_t *p = tmp.data() + (size_t)sim * 5 * B;  // [ix][iy][action][len][vtp]
            int32_t *o = h_out + ((size_t)sim * B + b) * 4;
            o[0] = p[b]; o[1] = p[2 * B + b]; o[2] = p[3 * B + b]; o[3] = p[4 * B + b];
        }
    return LZ_OK;
}

extern "C" int lz_roots_get_node_depths(lz_roots *r, int num_simulations, int32_t *h_out)
{
    LZ_REQUIRE(r != nullptr && h_out != nullptr, "NULL argument");
    LZ_REQUIRE(num_simulations >= 1 && num_simulations < r->t.NN, "num_simulations out of range");
    LZ_REQUIRE(r->t.variant == LZ_TREE_EFFICIENTZERO || r->t.variant == LZ_TREE_MUZERO, "node depths are kept by the EfficientZero / MuZero tree only");
    const size_t B = r->t.B, NN = r->t.NN;
    std::vector<uint64_t> tmp(B * NN);
    hipStream_t s = r->eng->stream;
    LZ_HIP_CHECK(hipMemcpyAsync(tmp.data(), r->t.node_link, tmp.size() * 8, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    for (size_t b = 0; b < B; ++b)
        for (int sim = 0; sim < num_simulations; ++sim)
            h_out[b * num_simulations + sim] = (int32_t)(tmp[b * NN + sim + 1] & 0xffffffull);
    return LZ_OK;
}

extern "C" int lz_roots_read_sim_outputs(lz_roots *r, int slot, float *h_value_prefix, float *h_value, float *h_policy_logits)
{
    LZ_REQUIRE(r != nullptr && r->pool_slab != nullptr, "no pools");
    LZ_REQUIRE(slot >= 0 && slot < r->t.NN, "slot out of range");
    const size_t B = r->t.B, A = policy_width_of(r->eng->model, r->t);
    hipStream_t s = r->eng->stream;
    if (h_value_prefix) LZ_HIP_CHECK(hipMemcpyAsync(h_value_prefix, r->sim_vp + slot * B, B * 4, hipMemcpyDeviceToHost, s));
    if (h_value) LZ_HIP_CHECK(hipMemcpyAsync(h_value, r->sim_value + slot * B, B * 4, hipMemcpyDeviceToHost, s));
    if (h_policy_logits) LZ_HIP_CHECK(hipMemcpyAsync(h_policy_logits, r->sim_logits + slot * B * A, B * A * 4, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    return LZ_OK;
}

extern "C" int lz_roots_read_latent(lz_roots *r, int slot, float *h_out_nchw)
{
    LZ_REQUIRE(r != nullptr && r->pool_slab != nullptr && h_out_nchw != nullptr, "no pools");
    LZ_REQUIRE(slot >= 0 && slot < r->t.NN, "slot out of range");
    lz_model *m = r->eng->model;
    const size_t B = r->t.B, C = m->cfg.num_channels, HW = m->HWl;  // MLP models: HW = 1, C = latent_state_dim
    std::vector<float> tmp(B * HW * C);
    hipStream_t s = r->eng->stream;
    LZ_HIP_CHECK(hipMemcpyAsync(tmp.data(), r->latent_pool + slot * B * HW * C, tmp.size() * 4, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    for (size_t b = 0; b < B; ++b)
        for (size_t p = 0; p < HW; ++p)
            for (size_t ch = 0; ch < C; ++ch) h_out_nchw[(b * C + ch) * HW + p] = tmp[(b * HW + p) * C + ch];
    return LZ_OK;
}

extern "C" int lz_roots_read_hidden(lz_roots *r, int slot, float *h_h, float *h_c)
{
    LZ_REQUIRE(r != nullptr && r->pool_slab != nullptr, "no pools");
    LZ_REQUIRE(slot >= 0 && slot < r->t.NN, "slot out of range");
    const size_t B = r->t.B, H = r->eng->model->cfg.lstm_hidden_size;
    hipStream_t s = r->eng->stream;
    if (h_h) LZ_HIP_CHECK(hipMemcpyAsync(h_h, r->h_pool + slot * B * H, B * H * 4, hipMemcpyDeviceToHost, s));
    if (h_c) LZ_HIP_CHECK(hipMemcpyAsync(h_c, r->c_pool + slot * B * H, B * H * 4, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    return LZ_OK;
}

// ---- teacher-forced / foreign-driver access to the pools
extern "C" int lz_roots_write_latent(lz_roots *r, int slot, const float *h_in_nchw)
{
    LZ_REQUIRE(r != nullptr && h_in_nchw != nullptr, "NULL argument");
    lz_model *m = r->eng->model;
    LZ_REQUIRE(m != nullptr && m->finalized, "no finalized model on this engine");
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    int rc = m->cfg.model_type >= 2 ? lz_mlp_ensure_pools(r) : ensure_pools(r);
    if (rc != LZ_OK) return rc;
    LZ_REQUIRE(slot >= 0 && slot < r->t.NN, "slot out of range");
    const size_t B = r->t.B, C = m->cfg.num_channels, HW = m->HWl;
    std::vector<float> tmp(B * HW * C);
    for (size_t b = 0; b < B; ++b)
        for (size_t p = 0; p < HW; ++p)
            for (size_t ch = 0; ch < C; ++ch) tmp[(b * HW + p) * C + ch] = h_in_nchw[(b * C + ch) * HW + p];
    hipStream_t s = r->eng->stream;
    LZ_HIP_CHECK(hipMemcpyAsync(r->latent_pool + slot * B * HW * C, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    r->inferred = true;  // the pools now hold a caller-provided state
    return LZ_OK;
}

extern "C" int lz_roots_write_hidden(lz_roots *r, int slot, const float *h_h, const float *h_c)
{
    LZ_REQUIRE(r != nullptr && h_h != nullptr && h_c != nullptr, "NULL argument");
    lz_model *m = r->eng->model;
    LZ_REQUIRE(m != nullptr && m->finalized, "no finalized model on this engine");
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    int rc = m->cfg.model_type >= 2 ? lz_mlp_ensure_pools(r) : ensure_pools(r);
    if (rc != LZ_OK) return rc;
    LZ_REQUIRE(slot >= 0 && slot < r->t.NN, "slot out of range");
    const size_t B = r->t.B, H = m->cfg.model_type >= 2 ? (size_t)lz_mlp_hidden_size(m) : (size_t)(m->cfg.model_type == 0 ? m->cfg.lstm_hidden_size : 0);
    LZ_REQUIRE(H > 0, "this model has no LSTM state");
    hipStream_t s = r->eng->stream;
    LZ_HIP_CHECK(hipMemcpyAsync(r->h_pool + slot * B * H, h_h, B * H * 4, hipMemcpyHostToDevice, s));
    LZ_HIP_CHECK(hipMemcpyAsync(r->c_pool + slot * B * H, h_c, B * H * 4, hipMemcpyHostToDevice, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    return LZ_OK;
}

// Model.recurrent_inference(latent_state, reward_hidden_state, action) (efficientzero_model.py:240-273, muzero_model.py:240-272)
// on pool slots: root i reads slot h_parent_slot[i], the results land in out_slot (read them with lz_roots_read_*).
extern "C" int lz_recurrent_inference(lz_roots *r, const int32_t *h_parent_slot, const int32_t *h_actions, const float *h_actions_f,
                                      const int32_t *h_search_len, int lstm_horizon_len, int out_slot)
{
    LZ_REQUIRE(r != nullptr && h_parent_slot != nullptr, "NULL argument");
    lz_model *m = r->eng->model;
    LZ_REQUIRE(m != nullptr && m->finalized, "no finalized model on this engine");
    LZ_REQUIRE(r->pool_slab != nullptr && r->pool_model_uid == r->eng->model_uid && r->inferred,
               "the pools hold no state: run lz_initial_inference (or lz_roots_write_latent) first");
    LZ_REQUIRE(out_slot >= 1 && out_slot < r->t.NN, "out_slot out of range");
    const lz_tree_dev &t = r->t;
    const size_t B = t.B;
    const bool sampled = t.variant == LZ_TREE_SAMPLED_EFFICIENTZERO;
    LZ_REQUIRE(sampled ? h_actions_f != nullptr : h_actions != nullptr, "actions: int32 [B] (float [B][D] for sampled roots)");
    for (size_t i = 0; i < B; ++i) {
        LZ_REQUIRE(h_parent_slot[i] >= 0 && h_parent_slot[i] < t.NN && h_parent_slot[i] != out_slot, "parent slot out of range");
        if (!sampled) LZ_REQUIRE(h_actions[i] >= 0 && h_actions[i] < m->cfg.action_space_size, "action out of range");
    }
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    hipStream_t s = r->eng->stream;
    std::vector<int32_t> ones(B, 1);
    LZ_HIP_CHECK(hipMemcpyAsync(t.res_ix, h_parent_slot, B * 4, hipMemcpyHostToDevice, s));
    if (!sampled) LZ_HIP_CHECK(hipMemcpyAsync(t.res_last_action, h_actions, B * 4, hipMemcpyHostToDevice, s));
    else LZ_HIP_CHECK(hipMemcpyAsync(t.res_last_action_f, h_actions_f, B * (size_t)t.D * 4, hipMemcpyHostToDevice, s));
    std::vector<int32_t> disc_idx;
    if (sampled && t.disc_A > 0) {  // discrete sampled roots: an action is the float of its index; the network encodes the index
        disc_idx.resize(B);
        for (size_t i = 0; i < B; ++i) {
            disc_idx[i] = (int32_t)h_actions_f[i];
            LZ_REQUIRE(disc_idx[i] >= 0 && disc_idx[i] < t.disc_A, "action out of range");
        }
        LZ_HIP_CHECK(hipMemcpyAsync(t.res_last_action, disc_idx.data(), B * 4, hipMemcpyHostToDevice, s));
    }
    LZ_HIP_CHECK(hipMemcpyAsync(t.res_search_len, h_search_len ? h_search_len : ones.data(), B * 4, hipMemcpyHostToDevice, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));  // the sources are caller / stack memory
    recurrent(r, out_slot - 1, h_search_len ? lstm_horizon_len : 0, s);
    LZ_HIP_CHECK(hipGetLastError());
    return LZ_OK;
}

extern "C" int lz_roots_read_debug_logits(lz_roots *r, int which, float *h_out)
{
    LZ_REQUIRE(r != nullptr && r->pool_slab != nullptr && h_out != nullptr && (which == 0 || which == 1), "bad argument");
    LZ_REQUIRE(r->trace_on, "the debug logits are written only while tracing is on (lz_roots_enable_trace before the inference)");
    const lz_model_cfg &mc = r->eng->model->cfg;
    const size_t B = r->t.B, SUP = (which == 1 && mc.reward_support_size > 0) ? mc.reward_support_size : mc.support_size;
    hipStream_t s = r->eng->stream;
    LZ_HIP_CHECK(hipMemcpyAsync(h_out, r->dbg_logits[which], B * SUP * 4, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    return LZ_OK;
}

// ---- debugging aids (not part of the documented ABI surface used by the shim)
extern "C" int lz_debug_set(lz_engine *e, const char *key, int value)
{
    LZ_REQUIRE(e && e->model && key, "bad argument");
    if (strcmp(key, "stop_stage") == 0) { e->model->debug_stop = value; return LZ_OK; }
    lz_set_error("unknown debug key %s", key);
    return LZ_ERR_INVALID;
}
extern "C" int lz_debug_read_ws(lz_engine *e, int which, float *h_out, int64_t n)
{
    LZ_REQUIRE(e && e->model && which >= 0 && which < 3 && e->model->ws[which] && h_out, "bad argument");
    LZ_HIP_CHECK(hipDeviceSynchronize());
    LZ_HIP_CHECK(hipMemcpy(h_out, e->model->ws[which], (size_t)n * 4, hipMemcpyDeviceToHost));
    return LZ_OK;
}

extern "C" int lz_debug_read_param(lz_engine *e, const char *name, float *h_out, int64_t n)
{
    LZ_REQUIRE(e && e->model && name && h_out, "bad argument");
    lz_model *m = e->model;
    const float *src = nullptr;
    if (!strcmp(name, "first_w")) src = m->first_w;
    else if (!strcmp(name, "first_s")) src = m->first_s;
    else if (!strcmp(name, "first_t")) src = m->first_t;
    else if (!strcmp(name, "r1a_w")) src = m->r1a.w;
    else if (!strcmp(name, "r1a_s")) src = m->r1a.scale;
    else if (!strcmp(name, "r1a_t")) src = m->r1a.shift;
    LZ_REQUIRE(src != nullptr, "unknown param");
    LZ_HIP_CHECK(hipMemcpy(h_out, src, (size_t)n * 4, hipMemcpyDeviceToHost));
    return LZ_OK;
}

// ---- in-stream kernel timing for the roofline line of bench.py: the tagged kernel class is the plain
// 64->64 3x3 convolution on the 6x6 latent (dynamics + prediction residual blocks, 4 launches / simulation).
extern "C" int lz_profile_enable(lz_engine *e, int max_launches)
{
    LZ_REQUIRE(e != nullptr && max_launches >= 0, "bad argument");
    LZ_HIP_CHECK(hipSetDevice(e->device));
    for (hipEvent_t ev : e->prof_ev) (void)hipEventDestroy(ev);
    e->prof_ev.clear();
    e->prof_used = 0;
    e->prof_on = max_launches > 0;
    for (int i = 0; i < 2 * max_launches; ++i) {
        hipEvent_t ev;
        LZ_HIP_CHECK(hipEventCreate(&ev));
        e->prof_ev.push_back(ev);
    }
    return LZ_OK;
}
extern "C" int lz_profile_read(lz_engine *e, int64_t *out_launches, double *out_total_ms)
{
    LZ_REQUIRE(e != nullptr && out_launches && out_total_ms, "bad argument");
    LZ_HIP_CHECK(hipStreamSynchronize(e->stream));
    double tot = 0.0;
    for (size_t i = 0; i < e->prof_used; ++i) {
        float ms = 0.f;
        LZ_HIP_CHECK(hipEventElapsedTime(&ms, e->prof_ev[2 * i], e->prof_ev[2 * i + 1]));
        tot += ms;
    }
    *out_launches = (int64_t)e->prof_used;
    *out_total_ms = tot;
    return LZ_OK;
}
