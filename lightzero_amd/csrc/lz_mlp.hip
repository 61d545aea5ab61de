// lz_mlp.hip -- the vector-observation (MLP) model family on the device:
//   MuZeroModelMLP                 lzero/model/muzero_model_mlp.py:13-338      (model_type 2, BASELINE configs[0])
//   EfficientZeroModelMLP          lzero/model/efficientzero_model_mlp.py      (model_type 3)
//   SampledEfficientZeroModelMLP   lzero/model/sampled_efficientzero_model_mlp.py (model_type 4, BASELINE configs[4])
// Weights are ingested by their state_dict names; every nn.Sequential produced by MLP_V2 (common.py:28-98) or
// ding.torch_utils.MLP is walked as Linear [, BatchNorm1d | LayerNorm] [, activation] groups.  The layers of one
// inference are levelised by their dependencies; every level is one k_dense launch (blockIdx.y = independent layer)
// plus, for the EfficientZero variants, the LSTM kernel.
#include <string.h>

#include <new>

#include "lz_model.h"
#include "lz_mlp.h"

struct DenseW {
    float *wf = nullptr, *bias = nullptr, *scale = nullptr, *shift = nullptr, *ln_g = nullptr, *ln_b = nullptr;
    int K = 0, N = 0, act = 0;
};

struct lz_mlp_model {
    int OBS = 0, L = 0, H = 0, A = 0, ENC = 0, PA = 0, SUP = 0, RSUP = 0, Wmax = 0;   // RSUP: the reward head's support (MuZeroModelMLP may have its own)
    float rsup_min = 0.0f;
    bool lstm = false, res = false, continuous = false, state_norm = false, scalar = false;   // state_norm / scalar: lz_model_cfg::state_norm / scalar_heads
    int enc_mode = 2;  // lz_dense_job.x2_mode of the action encoding
    std::vector<DenseW> rep, dyn1, dyn2, rew, common, val, pol;
    float *lstm_w = nullptr, *lstm_wf = nullptr, *lstm_b = nullptr;
};

void lz_mlp_model_destroy(lz_mlp_model *mm) { delete mm; }

namespace {

struct MlpBuilder {
    Builder &b;
    lz_model *m;
    int act_code;
    // [N][K] row-major -> MFMA-fragment order [Np/16][Kp/16][64][4]
    float *pack(const std::vector<float> &w, int N, int K)
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
        return b.upload(f);
    }
    const HostTensor *find(const std::string &name, size_t ndim)
    {
        auto it = m->raw.find(name);
        if (it == m->raw.end() || it->second.shape.size() != ndim) return nullptr;
        return &it->second;
    }
    DenseW linear(const std::string &lin, const std::string &norm, int act)
    {
        DenseW d;
        const HostTensor *w = find(lin + ".weight", 2);
        if (!w) { if (b.err.empty()) b.err = "missing tensor '" + lin + ".weight'"; return d; }
        d.N = (int)w->shape[0];
        d.K = (int)w->shape[1];
        const HostTensor *bias = b.get(lin + ".bias", {d.N});
        if (!bias) return d;
        d.wf = pack(w->data, d.N, d.K);
        d.bias = b.upload(bias->data);
        d.act = act;
        if (!norm.empty()) {
            if (find(norm + ".running_mean", 1)) {
                std::vector<float> sc, sh;
                b.bn(norm, d.N, sc, sh);
                d.scale = b.upload(sc);
                d.shift = b.upload(sh);
            } else {
                const HostTensor *g = b.get(norm + ".weight", {d.N}), *be = b.get(norm + ".bias", {d.N});
                if (g && be) { d.ln_g = b.upload(g->data); d.ln_b = b.upload(be->data); }
            }
        }
        return d;
    }
    // walks nn.Sequential(Linear [, norm] [, act], ..., Linear [, norm] [, act])
    std::vector<DenseW> seq(const std::string &prefix, int act, bool out_act)
    {
        std::vector<DenseW> out;
        std::vector<std::pair<int, int>> lin;  // (linear index, norm index or -1)
        for (int idx = 0; idx < 32; ++idx) {
            if (!find(prefix + "." + std::to_string(idx) + ".weight", 2)) continue;
            int norm = -1;
            if (find(prefix + "." + std::to_string(idx + 1) + ".weight", 1)) norm = idx + 1;
            lin.push_back({idx, norm});
        }
        if (lin.empty() && b.err.empty()) b.err = "no Linear layers under '" + prefix + "'";
        for (size_t i = 0; i < lin.size(); ++i) {
            const bool last = i + 1 == lin.size();
            out.push_back(linear(prefix + "." + std::to_string(lin[i].first),
                                 lin[i].second >= 0 ? prefix + "." + std::to_string(lin[i].second) : std::string(),
                                 (!last || out_act) ? act : 0));
        }
        return out;
    }
};

int widest(const std::vector<DenseW> &v, int w)
{
    for (const DenseW &d : v) if (d.N > w) w = d.N;
    return w;
}

}  // namespace

int lz_mlp_finalize(lz_engine *e)
{
    lz_model *m = e->model;
    const lz_model_cfg &c = m->cfg;
    delete m->mlp;
    m->mlp = new (std::nothrow) lz_mlp_model();
    if (!m->mlp) { lz_set_error("out of host memory"); return LZ_ERR_NOMEM; }
    lz_mlp_model &M = *m->mlp;
    Builder b{m, ""};
    const int act = c.activation == 1 ? 2 : 1;
    MlpBuilder mb{b, m, act};
    M.OBS = c.obs_c; M.L = c.num_channels; M.A = c.action_space_size; M.SUP = c.support_size;
    M.RSUP = c.reward_support_size > 0 ? c.reward_support_size : c.support_size;   // (lz_model_create admits it for model_type 2 only)
    M.rsup_min = c.reward_support_size > 0 ? c.reward_support_min : c.support_min;
    M.lstm = c.model_type != 2;
    M.H = M.lstm ? c.lstm_hidden_size : 0;
    M.res = c.res_connection_in_dynamics != 0;
    M.continuous = c.model_type == 4 && c.action_encoding == 2;
    M.state_norm = c.state_norm != 0; M.scalar = c.scalar_heads != 0;
    M.ENC = c.action_encoding == 1 ? 1 : M.A;
    M.enc_mode = c.action_encoding == 2 ? 1 : (c.action_encoding == 1 ? 3 : 2);
    M.PA = M.continuous ? 2 * M.A : M.A;
    // representation: MLP_V2 with the encoder's activation (GELU(tanh) unless the model passes its own:
    // common.py:803, muzero_model_mlp.py:108-110) and the final LayerNorm (common.py:838-839) folded into its last layer
    {
        MlpBuilder rb{b, m, c.model_type == 4 ? act : 2};
        M.rep = rb.seq("representation_network.fc_representation", rb.act_code, false);
        const HostTensor *g = b.get("representation_network.norm.weight", {M.L}), *be = b.get("representation_network.norm.bias", {M.L});
        if (!M.rep.empty() && g && be) {
            DenseW &last = M.rep.back();
            if (last.scale || last.ln_g) { if (b.err.empty()) b.err = "unexpected norm on the last representation layer"; }
            last.ln_g = b.upload(g->data);
            last.ln_b = b.upload(be->data);
        }
    }
    const std::string d = "dynamics_network.", p = "prediction_network.";
    if (M.res) {
        M.dyn1 = mb.seq(d + "fc_dynamics_1", act, true);
        M.dyn2 = mb.seq(d + "fc_dynamics_2", act, true);
    } else {
        M.dyn1 = mb.seq(d + "fc_dynamics", act, true);
    }
    M.rew = mb.seq(d + "fc_reward_head", act, false);
    M.common = mb.seq(p + "fc_prediction_common", act, true);
    M.val = mb.seq(p + "fc_value_head", act, false);
    if (M.continuous) {
        // ReparameterizationHead: main = Linear-ReLU-Linear-ReLU, then mu | log_sigma as ONE layer of width 2 D
        M.pol = mb.seq(p + "fc_policy_head.main", 1, true);
        const HostTensor *wm = b.get(p + "fc_policy_head.mu.weight", {M.A, M.L}), *bm = b.get(p + "fc_policy_head.mu.bias", {M.A});
        if (c.sigma_type != 0) { if (b.err.empty()) b.err = "only sigma_type 'conditioned' is compiled"; }
        const HostTensor *ws = b.get(p + "fc_policy_head.log_sigma_layer.weight", {M.A, M.L}), *bs = b.get(p + "fc_policy_head.log_sigma_layer.bias", {M.A});
        if (wm && bm && ws && bs) {
            std::vector<float> w(wm->data), bb(bm->data);
            w.insert(w.end(), ws->data.begin(), ws->data.end());
            bb.insert(bb.end(), bs->data.begin(), bs->data.end());
            DenseW ms;
            ms.N = 2 * M.A; ms.K = M.L; ms.act = 0;
            ms.wf = mb.pack(w, ms.N, ms.K);
            ms.bias = b.upload(bb);
            M.pol.push_back(ms);
        }
    } else {
        M.pol = mb.seq(p + "fc_policy_head", act, false);
    }
    if (M.lstm) {
        const int H = M.H, KX = M.L, K = KX + H;
        const HostTensor *wih = b.get(d + "lstm.weight_ih_l0", {4 * H, KX}), *whh = b.get(d + "lstm.weight_hh_l0", {4 * H, H}),
                         *bih = b.get(d + "lstm.bias_ih_l0", {4 * H}), *bhh = b.get(d + "lstm.bias_hh_l0", {4 * H});
        if (wih && whh && bih && bhh) {
            std::vector<float> wc((size_t)4 * H * K), bc((size_t)4 * H);
            for (int g = 0; g < 4; ++g)
                for (int u = 0; u < H; ++u) {
                    const int src = g * H + u, dst = 4 * u + g;
                    for (int k = 0; k < KX; ++k) wc[(size_t)dst * K + k] = wih->data[(size_t)src * KX + k];
                    for (int k = 0; k < H; ++k) wc[(size_t)dst * K + KX + k] = whh->data[(size_t)src * H + k];
                    bc[dst] = bih->data[src] + bhh->data[src];
                }
            M.lstm_w = b.upload(wc);
            M.lstm_b = b.upload(bc);
            if (K % 16 == 0 && H % 16 == 0) {
                std::vector<float> wf(wc.size());
                lz_lstm_pack_fragments(wc.data(), H, K, wf.data());
                M.lstm_wf = b.upload(wf);
            }
        }
    }
    if (b.err.empty()) {
        // shape checks the kernels rely on
        if (M.rep.empty() || M.rep.front().K != M.OBS || M.rep.back().N != M.L) b.err = "representation network shape mismatch";
        else if (M.dyn1.empty() || M.dyn1.front().K != M.L + M.ENC || M.dyn1.back().N != M.L) b.err = "dynamics network shape mismatch (latent + action encoding)";
        else if (M.res && (M.dyn2.empty() || M.dyn2.back().N != M.L)) b.err = "fc_dynamics_2 shape mismatch";
        else if (M.rew.empty() || M.rew.back().N != M.RSUP || M.rew.front().K != (M.lstm ? M.H : M.L)) b.err = "reward head shape mismatch";
        else if (M.val.empty() || M.val.back().N != M.SUP) b.err = "value head shape mismatch";
        else if (M.pol.empty() || M.pol.back().N != M.PA) b.err = "policy head shape mismatch";
        else if (M.common.empty() || M.common.back().N != M.L) b.err = "fc_prediction_common shape mismatch";
    }
    for (const auto *v : {&M.rep, &M.dyn1, &M.dyn2, &M.rew, &M.common, &M.val, &M.pol})
        for (const DenseW &dw : *v) {
            const bool first = v == &M.rep && &dw == &M.rep.front();   // raw observations may be wide (k_dense_wide: MiniGrid's 2835 features)
            if (b.err.empty() && ((dw.K > 640 && !(first && dw.K <= 16384)) || dw.N > 640))
                b.err = "dense layer beyond the compiled limits (in_features <= 640 -- the observation: <= 16384 --, out_features <= 640)";
        }
    if (b.err.empty() && M.lstm) {
        // the LSTM input may carry a deferred LayerNorm / activation, which only the k_lstm2 instantiations below apply
        const std::vector<DenseW> &src = M.res ? M.dyn2 : M.dyn1;
        const bool deferred = !src.empty() && src.back().ln_g;
        const bool compiled = (M.L == 256 && M.H == 512) || (M.L == 256 && M.H == 256) || (M.L == 128 && M.H == 128);
        if (deferred && !compiled) b.err = "LayerNorm MLP models are compiled for (latent_state_dim, lstm_hidden_size) = (256, 512), (256, 256) and (128, 128)";
    }
    if (!b.err.empty()) { lz_set_error("lz_model_finalize: %s", b.err.c_str()); return LZ_ERR_STATE; }
    int w = std::max(M.L, M.H);
    for (const auto *v : {&M.rep, &M.dyn1, &M.dyn2, &M.rew, &M.common, &M.val, &M.pol}) w = widest(*v, w);
    M.Wmax = w;
    LZ_HIP_CHECK(hipDeviceSynchronize());  // weight uploads went through the null stream; the engine stream does not order against it
    m->finalized = true;
    return LZ_OK;
}

// ------------------------------------------------------------------------------------------------
static size_t align_up_(size_t x, size_t a) { return (x + a - 1) / a * a; }

int lz_roots_release_pools_if_stale(lz_roots *r);  // lz_search.hip

int lz_mlp_ensure_pools(lz_roots *r)
{
    if (int rc = lz_roots_release_pools_if_stale(r)) return rc;
    if (r->pool_slab) return LZ_OK;
    const lz_mlp_model &M = *r->eng->model->mlp;
    const size_t B = r->t.B, NN = r->t.NN, L = M.L, H = M.H, PA = M.PA, W = M.Wmax, SUP = std::max(M.SUP, M.RSUP);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up_(off + bytes, 256); return o; };
    const size_t o_lat = take(NN * B * L * 4), o_h = take(NN * B * H * 4), o_c = take(NN * B * H * 4), o_vp = take(NN * B * 4),
                 o_val = take(NN * B * 4), o_lg = take(NN * B * PA * 4), o_d0 = take(B * SUP * 4), o_d1 = take(B * SUP * 4),
                 o_tr = take(NN * 5 * B * 4), o_z = take(B * 4),
                 o_tp = take((2 * B + B * std::max<size_t>(PA, r->t.A)) * 4);   // [to_play | noise offsets | noise]: one upload per prepare
    size_t o_mt[14];
    for (int i = 0; i < 14; ++i) o_mt[i] = take(B * W * 4);
    hipError_t err = lz_dev_malloc((void **)&r->pool_slab, off);
    if (err != hipSuccess) {
        lz_set_error("hipMalloc(%zu bytes) for the latent/LSTM pools failed: %s", off, hipGetErrorString(err));
        return err == hipErrorOutOfMemory ? LZ_ERR_NOMEM : LZ_ERR_HIP;
    }
    char *base = (char *)r->pool_slab;
    r->latent_pool = (float *)(base + o_lat); r->h_pool = (float *)(base + o_h); r->c_pool = (float *)(base + o_c);
    r->sim_vp = (float *)(base + o_vp); r->sim_value = (float *)(base + o_val); r->sim_logits = (float *)(base + o_lg);
    r->dbg_logits[0] = (float *)(base + o_d0); r->dbg_logits[1] = (float *)(base + o_d1);
    r->trace = (int32_t *)(base + o_tr); r->d_to_play = (int32_t *)(base + o_tp); r->d_zero_vp = (float *)(base + o_z);
    r->d_noise_off = r->d_to_play + B; r->d_noise = (float *)(r->d_noise_off + B);
    for (int i = 0; i < 14; ++i) r->mt[i] = (float *)(base + o_mt[i]);
    r->t_hbn = r->mt[13];
    LZ_HIP_CHECK(hipMemsetAsync(r->d_zero_vp, 0, B * 4, r->eng->stream));
    return LZ_OK;
}

namespace {

// a [B][K] activation in HBM plus the transform its consumers apply while loading it (the producer's deferred
// LayerNorm + activation, then an optional residual); `materialise`: where one consumer writes the transformed rows
struct Act {
    const float *x = nullptr;
    const int32_t *gather = nullptr;
    int64_t slot_stride = 0;
    int K = 0;
    const float *ln_g = nullptr, *ln_b = nullptr;
    int act = 0;
    const float *res = nullptr;
    const int32_t *res_gather = nullptr;
    int64_t res_slot_stride = 0;
    float *materialise = nullptr;
    bool minmax = false;   // state_norm=True: the consumer renormalises the transformed row over its K columns (lz_dense_job::in_minmax)
};

// one inference as dependency levels of dense jobs (+ row finishers, + the LSTM after the jobs of `lstm_level`)
struct Program {
    std::vector<std::vector<lz_dense_job>> levels;
    std::vector<std::vector<lz_rowfinal_job>> finals;
    int lstm_level = -1;
    lz_lstm_args lstm{};
    float ln_eps = 1e-5f;
    lz_dense_job &add(int level, const lz_dense_job &j)
    {
        if ((int)levels.size() <= level) levels.resize(level + 1);
        levels[level].push_back(j);
        return levels[level].back();
    }
    void add_final(int level, const lz_rowfinal_job &j)
    {
        if ((int)finals.size() <= level) finals.resize(level + 1);
        finals[level].push_back(j);
    }
    void run(int B, hipStream_t s)
    {
        const int n = std::max(std::max((int)levels.size(), (int)finals.size()), lstm_level + 1);
        for (int l = 0; l < n; ++l) {
            if (l < (int)levels.size()) {
                const auto &v = levels[l];
                for (size_t i = 0; i < v.size(); i += 4) {
                    lz_dense_args a{};
                    a.B = B;
                    a.njobs = (int)std::min<size_t>(4, v.size() - i);
                    for (int k = 0; k < a.njobs; ++k) a.job[k] = v[i + k];
                    lz_launch_dense(a, s);
                }
            }
            if (l < (int)finals.size() && !finals[l].empty()) {
                lz_rowfinal_args a{};
                a.B = B;
                a.njobs = (int)std::min<size_t>(4, finals[l].size());
                for (int k = 0; k < a.njobs; ++k) a.job[k] = finals[l][k];
                lz_launch_rowfinal(a, s);
            }
            if (l == lstm_level) lz_launch_lstm(lstm, s);
        }
    }
    // layer `w` reading `in` at `level`; returns the job (already added) -- its output activation is out_of(job, w)
    lz_dense_job &layer(int level, const DenseW &w, Act &in, float *out)
    {
        lz_dense_job j{};
        j.x = in.x; j.x_gather = in.gather; j.x_slot_stride = in.slot_stride; j.K1 = in.K;
        j.in_ln_g = in.ln_g; j.in_ln_b = in.ln_b; j.in_ln_eps = ln_eps; j.in_act = in.act;
        j.in_res = in.res; j.in_res_gather = in.res_gather; j.in_res_slot_stride = in.res_slot_stride;
        j.in_out = in.materialise;
        j.in_minmax = in.minmax ? 1 : 0;
        in.materialise = nullptr;  // one writer is enough
        j.wf = w.wf; j.bias = w.bias; j.scale = w.scale; j.shift = w.shift; j.N = w.N;
        j.act = w.ln_g ? 0 : w.act;  // with a LayerNorm the activation is deferred to the consumers together with it
        j.out = out;
        return add(level, j);
    }
    static Act out_of(const lz_dense_job &j, const DenseW &w)
    {
        Act o;
        o.x = j.out; o.K = w.N;
        if (w.ln_g) { o.ln_g = w.ln_g; o.ln_b = w.ln_b; o.act = w.act; }
        return o;
    }
    // layers as a chain from `level`; intermediate activations alternate between t0 / t1 (the last one lands in `last_out`
    // when given).  Returns the level of the last layer; `out` = the chain's output activation.
    int chain(int level, const std::vector<DenseW> &layers, Act in, float *t0, float *t1, float *last_out, Act *out,
              lz_dense_job **last_job = nullptr)
    {
        Act cur = in;
        for (size_t i = 0; i < layers.size(); ++i) {
            const bool last = i + 1 == layers.size();
            float *dst = (last && last_out) ? last_out : ((i & 1) ? t1 : t0);
            lz_dense_job &j = layer(level, layers[i], cur, dst);
            if (last && last_job) *last_job = &j;
            cur = out_of(j, layers[i]);
            ++level;
        }
        if (out) *out = cur;
        return level - 1;
    }
};

// prediction network on `latent`: trunk, then the value chain (-> row finisher) and the policy chain off the trunk
// returns the level after which the value logits are complete; `value_final` = the row finisher the caller schedules
int prediction(Program &P, int level, const lz_mlp_model &M, lz_roots *r, Act &latent, float support_min, float *out_value,
               float *out_policy, float *value_logits, lz_rowfinal_job *value_final)
{
    Act pc;
    lz_dense_job *first = nullptr;
    // the first trunk layer may have to materialise its (transformed) input: chain() hands `latent` by value, so do layer 0 here
    {
        lz_dense_job &j = P.layer(level, M.common[0], latent, r->mt[4]);
        first = &j;
        pc = Program::out_of(j, M.common[0]);
        (void)first;
    }
    int lc = level;
    if (M.common.size() > 1) {
        std::vector<DenseW> rest(M.common.begin() + 1, M.common.end());
        lc = P.chain(level + 1, rest, pc, r->mt[5], r->mt[4], nullptr, &pc);
    }
    Act vout;
    const int lv = P.chain(lc + 1, M.val, pc, r->mt[6], r->mt[7], value_logits, &vout);
    value_final->logits = value_logits; value_final->N = M.SUP; value_final->support_min = support_min; value_final->out_scalar = out_value;
    value_final->scalar = M.scalar ? 1 : 0;
    lz_dense_job *pl = nullptr;
    P.chain(lc + 1, M.pol, pc, r->mt[8], r->mt[9], out_policy, nullptr, &pl);
    if (M.continuous) { pl->final = 2; pl->final_split = M.A; pl->final_tanh = r->eng->model->cfg.bound_type == 1; }
    return lv;
}

}  // namespace

int lz_mlp_initial_inference(lz_roots *r, const float *d_obs)
{
    lz_model *m = r->eng->model;
    const lz_mlp_model &M = *m->mlp;
    const lz_model_cfg &c = m->cfg;
    int rc = lz_mlp_ensure_pools(r);
    if (rc != LZ_OK) return rc;
    hipStream_t s = r->eng->stream;
    const int B = r->t.B;
    Program P;
    P.ln_eps = c.ln_eps > 0 ? c.ln_eps : 1e-5f;
    Act obs;
    obs.x = d_obs; obs.K = M.OBS;
    Act latent;
    const int lr = P.chain(0, M.rep, obs, r->mt[0], r->mt[1], r->mt[2], &latent);
    latent.materialise = r->latent_pool;  // slot 0: written by the first consumer with the encoder's final LayerNorm applied
    latent.minmax = M.state_norm;         // ... and, with state_norm=True, renormalised (muzero_model_mlp.py:220-221)
    lz_rowfinal_job vf{};
    const int lvv = prediction(P, lr + 1, M, r, latent, c.support_min, r->sim_value, r->sim_logits, r->dbg_logits[0], &vf);
    P.add_final(lvv + 1, vf);
    P.run(B, s);
    if (M.lstm) {
        LZ_HIP_CHECK(hipMemsetAsync(r->h_pool, 0, (size_t)B * M.H * 4, s));
        LZ_HIP_CHECK(hipMemsetAsync(r->c_pool, 0, (size_t)B * M.H * 4, s));
    }
    LZ_HIP_CHECK(hipGetLastError());
    r->inferred = true;
    r->inference_fresh = true;  // no prepare has consumed it yet (lz_roots_reset_keep_inference)
    return LZ_OK;
}

// recurrent_inference for the leaves of the last traverse, outputs into slot sim + 1 (muzero_model_mlp.py:212-238,
// efficientzero_model_mlp.py / sampled_efficientzero_model_mlp.py recurrent_inference)
void lz_mlp_recurrent(lz_roots *r, int sim, int horizon, hipStream_t s)
{
    lz_model *m = r->eng->model;
    const lz_mlp_model &M = *m->mlp;
    const lz_model_cfg &c = m->cfg;
    const lz_tree_dev &t = r->t;
    const size_t B = t.B;
    const int slot = sim + 1;
    const size_t lat_slot = B * M.L;
    float *next_latent = r->latent_pool + (size_t)slot * lat_slot;
    if (r->trace_on) (void)hipMemcpyAsync(r->trace + (size_t)sim * 5 * B, t.res_ix, 5 * B * 4, hipMemcpyDeviceToDevice, s);
    Program P;
    P.ln_eps = c.ln_eps > 0 ? c.ln_eps : 1e-5f;
    // dynamics trunk: [latent | action encoding] -> next latent (+ latent with res_connection_in_dynamics)
    Act z;
    z.x = r->latent_pool; z.gather = t.res_ix; z.slot_stride = (int64_t)lat_slot; z.K = M.L;
    Act next;
    const int lv = P.chain(0, M.dyn1, z, r->mt[0], r->mt[1], r->mt[2], &next);
    {
        lz_dense_job &first = P.levels[0][0];
        first.K2 = M.ENC; first.x2_mode = M.enc_mode;
        first.x2 = t.res_last_action_f; first.x2_idx = t.res_last_action; first.x2_div = (float)M.A;
    }
    if (M.res) { next.res = r->latent_pool; next.res_gather = t.res_ix; next.res_slot_stride = (int64_t)lat_slot; }
    // the pool must hold the finished next latent (deferred norm / activation / residual applied): its first consumer writes
    // it; without any deferred transform the trunk's last layer writes the pool slot itself
    // state_norm=True (muzero_model_mlp.py:291-295, efficientzero_model_mlp.py:307-308): what the prediction network and the pool get is the
    // renormalised next latent; the reward / value-prefix path below (`enc`) keeps the un-normalised one, like the reference's dynamics networks
    const bool deferred = next.ln_g || next.act || next.res || M.state_norm;
    if (deferred) next.materialise = next_latent;
    else {
        P.levels[lv].back().out = next_latent;
        next.x = next_latent;
    }
    // prediction first: it is the consumer that materialises the next latent
    Act enc = next;
    next.minmax = M.state_norm;
    lz_rowfinal_job vf{};
    const int lvv = prediction(P, lv + 1, M, r, next, c.support_min, r->sim_value + (size_t)slot * B,
                               r->sim_logits + (size_t)slot * B * M.PA, r->dbg_logits[0], &vf);
    enc.materialise = nullptr;
    int le = lv;
    if (M.res) le = P.chain(lv + 1, M.dyn2, enc, r->mt[10], r->mt[11], r->mt[3], &enc);
    // reward / value prefix
    Act rin = enc;
    if (M.lstm) {
        lz_lstm_args &l = P.lstm;
        l.x = enc.x; l.x_ln_g = enc.ln_g; l.x_ln_b = enc.ln_b; l.x_ln_eps = P.ln_eps; l.x_act = enc.act;
        l.h_pool = r->h_pool; l.c_pool = r->c_pool; l.gather_ix = t.res_ix; l.wcat = M.lstm_w; l.wf = M.lstm_wf; l.bias = M.lstm_b;
        l.bn_scale = nullptr; l.bn_shift = nullptr; l.search_len = t.res_search_len; l.horizon = horizon;
        l.h_out = r->h_pool + (size_t)slot * B * M.H; l.c_out = r->c_pool + (size_t)slot * B * M.H; l.hbn_out = r->t_hbn;
        l.B = (int)B; l.KX = M.L; l.H = M.H;
        P.lstm_level = le;  // after the dense jobs of the level that produced `enc`
        rin = Act{};
        rin.x = r->t_hbn; rin.K = M.H;
    }
    Act rout;
    const int lrw = P.chain(le + 1, M.rew, rin, r->mt[12], r->mt[1], r->dbg_logits[1], &rout);
    lz_rowfinal_job rf{};
    rf.logits = r->dbg_logits[1]; rf.N = M.RSUP; rf.support_min = M.rsup_min; rf.out_scalar = r->sim_vp + (size_t)slot * B;
    rf.scalar = M.scalar ? 1 : 0;
    // both row finishers (value, value prefix / reward) share the last launch
    const int lfin = std::max(lrw, lvv) + 1;
    P.add_final(lfin, vf);
    P.add_final(lfin, rf);
    P.run((int)B, s);
}

int lz_mlp_latent_size(const lz_model *m) { return m->mlp->L; }
int lz_mlp_hidden_size(const lz_model *m) { return m->mlp->H; }
int lz_mlp_policy_width(const lz_model *m) { return m->mlp->PA; }
int lz_mlp_obs_size(const lz_model *m) { return m->mlp->OBS; }
