// lz_capi.hip -- the extern "C" boundary (include/lz_mi355.h): handle lifetime, host<->HBM staging
// for the fine-grained tree API, error reporting.  No compute happens on the host.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <new>

#include <algorithm>
#include <cmath>
#include <random>
#include <vector>

#include <mutex>
#include <unordered_set>

#include "lz_internal.h"

static thread_local char g_err[1024] = "";

void lz_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *lz_last_error(void) { return g_err; }
extern "C" int lz_version(void) { return 100; }

extern "C" int lz_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ok++;
    }
    return ok;
}

void lz_model_destroy(lz_model *m);

extern "C" int lz_engine_create(int device_index, lz_engine **out)
{
    LZ_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        lz_set_error("no HIP device visible: liblz_mi355 has no CPU fallback");
        return LZ_ERR_NODEVICE;
    }
    LZ_REQUIRE(device_index >= 0 && device_index < n, "device_index out of range");
    hipDeviceProp_t p;
    LZ_HIP_CHECK(hipGetDeviceProperties(&p, device_index));
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        lz_set_error("device %d is %s; this library is built for gfx950 (MI355X) only", device_index, p.gcnArchName);
        return LZ_ERR_NODEVICE;
    }
    LZ_HIP_CHECK(hipSetDevice(device_index));
    lz_engine *e = new (std::nothrow) lz_engine();
    if (!e) { lz_set_error("out of host memory"); return LZ_ERR_NOMEM; }
    e->device = device_index;
    hipError_t err = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (err != hipSuccess) { delete e; lz_set_error("hipStreamCreate: %s", hipGetErrorString(err)); return LZ_ERR_HIP; }
    *out = e;
    return LZ_OK;
}

// Live roots handles.  A host-language binding frees handles from finalizers whose order it does not control (Python's garbage
// collector): destroying a roots handle twice, or after its engine, must be an error status, not a crash.  lz_roots_destroy therefore
// only accepts handles that are registered here, and lz_engine_destroy takes the engine's remaining roots down with it.
static std::mutex g_roots_mu;
static std::unordered_set<lz_roots *> g_live_roots;

extern "C" int lz_engine_destroy(lz_engine *e)
{
    if (!e) return LZ_OK;
    (void)hipSetDevice(e->device);
    std::vector<lz_roots *> mine;
    {
        std::lock_guard<std::mutex> lk(g_roots_mu);
        for (lz_roots *r : g_live_roots) if (r->eng == e) mine.push_back(r);
    }
    for (lz_roots *r : mine) (void)lz_roots_destroy(r);   // (a later lz_roots_destroy on one of them returns LZ_ERR_INVALID)
    if (e->model) lz_model_destroy(e->model);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return LZ_OK;
}

extern "C" int lz_engine_synchronize(lz_engine *e)
{
    LZ_REQUIRE(e != nullptr, "engine is NULL");
    LZ_HIP_CHECK(hipStreamSynchronize(e->stream));
    return LZ_OK;
}

extern "C" void *lz_engine_stream(lz_engine *e) { return e ? (void *)e->stream : nullptr; }

// ------------------------------------------------------------------------------------------------
static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static int ensure_stage(lz_roots *r, size_t bytes)
{
    // an upload that was left in flight (lz_roots_upload_legal) still reads the pinned buffer: every other user waits for it here
    if (r->stage_pending) {
        LZ_HIP_CHECK(hipEventSynchronize(r->stage_done));
        r->stage_pending = false;
    }
    if (bytes <= r->stage_bytes) return LZ_OK;
    if (r->h_stage) (void)hipHostFree(r->h_stage);
    if (r->d_stage) (void)hipFree(r->d_stage);
    r->h_stage = r->d_stage = nullptr;
    r->stage_bytes = 0;
    bytes = align_up(bytes, 4096);
    LZ_HIP_CHECK(hipHostMalloc(&r->h_stage, bytes, hipHostMallocDefault));
    LZ_HIP_CHECK(lz_dev_malloc((void **)&r->d_stage, bytes));
    r->stage_bytes = bytes;
    return LZ_OK;
}

int lz_roots_alloc(lz_engine *e, int variant, int B, int A, int max_sims, lz_roots **out, int D = 0)
{
    lz_roots *r = new (std::nothrow) lz_roots();
    if (!r) { lz_set_error("out of host memory"); return LZ_ERR_NOMEM; }
    r->eng = e;
    lz_tree_dev &t = r->t;
    t.B = B; t.A = A; t.NN = max_sims + 1; t.variant = variant; t.D = D; t.disc_A = 0;
    const size_t nBNA = (size_t)B * t.NN * A, nBN = (size_t)B * t.NN;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_edge = take(nBNA * sizeof(float4)), o_child = take(nBNA * 4), o_vp = take(nBN * 4),
                 o_reset = take(nBN * 4), o_tp = take(nBN * 4), o_best = take(nBN * 4), o_rv = take((size_t)B * 4),
                 o_rs = take((size_t)B * 4), o_legal = take((size_t)B * A * 4), o_nl = take((size_t)B * 4),
                 o_mm = take((size_t)B * 8), o_pn = take(nBN * 4), o_pa = take(nBN * 4), o_res = take((size_t)B * 4 * 5),
                 o_ep = take(256 + 512), o_rep = take(D ? nBNA * 4 : 0), o_nch = take(D ? nBN * 4 : 0),
                 o_act = take(D ? nBNA * D * 4 : 0), o_laf = take(D ? (size_t)B * D * 4 : 0),
                 o_bidx = take(nBN * 4), o_noinf = take((size_t)B * 4), o_link = take(nBN * 8),
                 o_raw = take(variant == LZ_TREE_GUMBEL_MUZERO ? nBN * 4 : 0), o_gum = take(variant == LZ_TREE_GUMBEL_MUZERO ? (size_t)A * 4 : 0),
                 o_cons = take(variant == LZ_TREE_GUMBEL_MUZERO ? (size_t)t.NN * 4 : 0),
                 o_gsoft = take(variant == LZ_TREE_GUMBEL_MUZERO ? nBNA * 4 : 0);
    hipError_t err = lz_dev_malloc((void **)&r->slab, off);
    if (err != hipSuccess) {
        delete r;
        lz_set_error("hipMalloc(%zu bytes) for the node pool failed: %s", off, hipGetErrorString(err));
        return err == hipErrorOutOfMemory ? LZ_ERR_NOMEM : LZ_ERR_HIP;
    }
    r->slab_bytes = off;
    char *base = (char *)r->slab;
    t.edge = (float4 *)(base + o_edge); t.child = (int32_t *)(base + o_child); t.node_vp = (float *)(base + o_vp);
    t.node_reset = (int32_t *)(base + o_reset); t.node_to_play = (int32_t *)(base + o_tp);
    t.node_best = (int32_t *)(base + o_best); t.root_visit = (int32_t *)(base + o_rv); t.root_vsum = (float *)(base + o_rs);
    t.legal = (int32_t *)(base + o_legal); t.n_legal = (int32_t *)(base + o_nl); t.minmax = (float *)(base + o_mm);
    t.path_node = (int32_t *)(base + o_pn); t.path_act = (int32_t *)(base + o_pa);
    int32_t *res = (int32_t *)(base + o_res);
    t.rep = D ? (int32_t *)(base + o_rep) : nullptr; t.nchild = D ? (int32_t *)(base + o_nch) : nullptr;
    t.actions = D ? (float *)(base + o_act) : nullptr; t.res_last_action_f = D ? (float *)(base + o_laf) : nullptr;
    t.rng_epoch = (uint32_t *)(base + o_ep);
    r->explore_tab = (float *)(base + o_ep + 256);
    r->tab_valid = false;
    t.node_bidx = (int32_t *)(base + o_bidx); t.res_noinf = (int32_t *)(base + o_noinf); t.node_link = (uint64_t *)(base + o_link);
    // no kernel may depend on what the allocator handed back (epoch, legal lists, results).  On the engine's own stream: that
    // stream is non-blocking, so a null-stream memset queued behind another library's work (torch's default stream) could land
    // AFTER the first prepare and wipe it.
    LZ_HIP_CHECK(hipMemsetAsync(r->slab, 0, off, e->stream));
    t.node_raw = nullptr; t.gumbel = nullptr; t.considered = nullptr; t.gsoft = nullptr;
    if (variant == LZ_TREE_GUMBEL_MUZERO) {
        t.node_raw = (float *)(base + o_raw); t.gumbel = (float *)(base + o_gum); t.considered = (int32_t *)(base + o_cons);
        t.gsoft = (float *)(base + o_gsoft);
        // every CNode draws its Gumbel vector from std::mt19937(gumbel_rng = 0) scaled by gumbel_scale = 10 (cnode.cpp:58-59,86-89,
        // 1133-1151): one constant prefix, computed with the same libstdc++ distribution the reference uses
        std::mt19937 gen(static_cast<unsigned int>(0.0f));
        std::extreme_value_distribution<float> dist(0, 1);
        std::vector<float> g((size_t)A);
        for (int i = 0; i < A; ++i) g[i] = 10.0f * dist(gen);
        LZ_HIP_CHECK(hipMemcpyAsync(t.gumbel, g.data(), (size_t)A * 4, hipMemcpyHostToDevice, e->stream));
        LZ_HIP_CHECK(hipStreamSynchronize(e->stream));  // g is a stack vector
    }
    t.res_ix = res; t.res_iy = res + B; t.res_last_action = res + 2 * B; t.res_search_len = res + 3 * B; t.res_vtp = res + 4 * B;
    {
        std::lock_guard<std::mutex> lk(g_roots_mu);
        g_live_roots.insert(r);
    }
    *out = r;
    return LZ_OK;
}

int lz_roots_upload_legal(lz_roots *r, const int32_t *h_legal_flat, const int32_t *h_legal_count)
{
    const lz_tree_dev &t = r->t;
    const int B = t.B, A = t.A;
    int rc = ensure_stage(r, (size_t)B * A * 4 + (size_t)B * 4);
    if (rc != LZ_OK) return rc;
    int32_t *hl = (int32_t *)r->h_stage;
    int32_t *hn = hl + (size_t)B * A;
    size_t off = 0;
    for (int i = 0; i < B; ++i) {
        int n = h_legal_count ? h_legal_count[i] : 0;
        if (n < 0 || n > A) { lz_set_error("legal action count %d of root %d out of range [0,%d]", n, i, A); return LZ_ERR_INVALID; }
        if (n == 0) {  // empty list == all actions (cnode.cpp:106-112)
            for (int a = 0; a < A; ++a) hl[(size_t)i * A + a] = a;
            hn[i] = A;
        } else {
            for (int j = 0; j < n; ++j) {
                int a = h_legal_flat[off + j];
                if (a < 0 || a >= A) { lz_set_error("legal action %d of root %d out of range [0,%d)", a, i, A); return LZ_ERR_INVALID; }
                hl[(size_t)i * A + j] = a;
            }
            for (int j = n; j < A; ++j) hl[(size_t)i * A + j] = 0;
            hn[i] = n;
            off += n;
        }
    }
    r->h_n_legal.assign(hn, hn + B);
    hipStream_t s = r->eng->stream;
    LZ_HIP_CHECK(hipMemcpyAsync(t.legal, hl, (size_t)B * A * 4, hipMemcpyHostToDevice, s));
    LZ_HIP_CHECK(hipMemcpyAsync(t.n_legal, hn, (size_t)B * 4, hipMemcpyHostToDevice, s));
    // No synchronisation: the policy surface re-arms the roots (reset_mask) right after launching the initial inference, and a wait
    // here parked the host for the whole representation tower (0.53 ms per collect step, tools/prof_policy.py) before it could enqueue
    // the prepare and the search.  The pinned buffer is protected by an event instead (ensure_stage).
    if (!r->stage_done) LZ_HIP_CHECK(hipEventCreateWithFlags(&r->stage_done, hipEventDisableTiming));
    LZ_HIP_CHECK(hipEventRecord(r->stage_done, s));
    r->stage_pending = true;
    return LZ_OK;
}

extern "C" int lz_roots_create(lz_engine *e, int variant, int root_num, int action_space_size, int max_simulations,
                               const int32_t *h_legal_flat, const int32_t *h_legal_count, lz_roots **out)
{
    LZ_REQUIRE(e != nullptr && out != nullptr, "engine/out is NULL");
    *out = nullptr;
    LZ_REQUIRE(variant == LZ_TREE_EFFICIENTZERO || variant == LZ_TREE_MUZERO || variant == LZ_TREE_GUMBEL_MUZERO, "unknown tree variant");
    LZ_REQUIRE(root_num > 0 && action_space_size > 0 && max_simulations > 0, "root_num, action_space_size, max_simulations must be positive");
    // beyond 256 actions (Chinese chess: 2086) the MuZero / EfficientZero trees run lz_tree_wide.hip; node_link keeps the action in 16 bits
    LZ_REQUIRE(action_space_size <= 65535, "action_space_size > 65535 is not supported");
    LZ_REQUIRE(action_space_size <= 1024 || variant != LZ_TREE_GUMBEL_MUZERO, "action_space_size > 1024 is not supported by the Gumbel MuZero tree kernels (16 register chunks per node)");
    LZ_HIP_CHECK(hipSetDevice(e->device));
    lz_roots *r = nullptr;
    int rc = lz_roots_alloc(e, variant, root_num, action_space_size, max_simulations, &r);
    if (rc != LZ_OK) return rc;
    rc = lz_roots_upload_legal(r, h_legal_flat, h_legal_count);
    if (rc != LZ_OK) { lz_roots_destroy(r); return rc; }
    lz_tree_launch_minmax_reset(r->t, e->stream);
    *out = r;
    return LZ_OK;
}

static int roots_rearm(lz_roots *r, const int32_t *h_legal_flat, const int32_t *h_legal_count, bool keep_inference)
{
    LZ_REQUIRE(r != nullptr, "roots is NULL");
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    int rc = lz_roots_upload_legal(r, h_legal_flat, h_legal_count);
    if (rc != LZ_OK) return rc;
    r->prepared = false;
    if (!keep_inference) { r->inferred = false; r->inference_fresh = false; }
    r->traverse_count = 0;
    lz_tree_launch_minmax_reset(r->t, r->eng->stream);
    return LZ_OK;
}
extern "C" int lz_roots_reset(lz_roots *r, const int32_t *h_legal_flat, const int32_t *h_legal_count)
{
    return roots_rearm(r, h_legal_flat, h_legal_count, false);
}
extern "C" int lz_roots_reset_keep_inference(lz_roots *r, const int32_t *h_legal_flat, const int32_t *h_legal_count)
{
    LZ_REQUIRE(r != nullptr && r->inferred && r->inference_fresh, "lz_roots_reset_keep_inference needs lz_initial_inference for this env-step first");
    return roots_rearm(r, h_legal_flat, h_legal_count, true);
}

extern "C" int lz_roots_destroy(lz_roots *r)
{
    if (!r) return LZ_OK;
    {
        std::lock_guard<std::mutex> lk(g_roots_mu);
        if (!g_live_roots.erase(r)) {
            lz_set_error("lz_roots_destroy: %p is not a live roots handle (destroyed before, or its engine was destroyed)", (void *)r);
            return LZ_ERR_INVALID;
        }
    }
    (void)hipSetDevice(r->eng->device);
    (void)hipStreamSynchronize(r->eng->stream);
    if (r->slab) (void)hipFree(r->slab);
    if (r->graph_exec) (void)hipGraphExecDestroy(r->graph_exec);
    if (r->pool_slab) (void)hipFree(r->pool_slab);
    if (r->hd_logits) (void)hipFree(r->hd_logits);
    if (r->stamps) (void)hipFree(r->stamps);
    if (r->d_obs) (void)hipFree(r->d_obs);
    if (r->d_given) (void)hipFree(r->d_given);
    if (r->h_prep) (void)hipHostFree(r->h_prep);
    if (r->d_results) (void)hipFree(r->d_results);
    if (r->h_results) (void)hipHostFree(r->h_results);
    if (r->prep_done) (void)hipEventDestroy(r->prep_done);
    if (r->stage_done) (void)hipEventDestroy(r->stage_done);
    if (r->rows_done) (void)hipEventDestroy(r->rows_done);
    if (r->d_reuse) (void)hipFree(r->d_reuse);
    if (r->h_stage) (void)hipHostFree(r->h_stage);
    if (r->d_stage) (void)hipFree(r->d_stage);
    delete r;
    return LZ_OK;
}

extern "C" int lz_roots_num(const lz_roots *r) { return r ? r->t.B : LZ_ERR_INVALID; }

extern "C" int lz_roots_minmax_reset(lz_roots *r, float value_delta_max)
{
    LZ_REQUIRE(r != nullptr, "roots is NULL");
    r->delta = value_delta_max;
    lz_tree_launch_minmax_reset(r->t, r->eng->stream);
    LZ_HIP_CHECK(hipGetLastError());
    return LZ_OK;
}

extern "C" int lz_roots_set_tiebreak(lz_roots *r, int mode, uint64_t seed)
{
    LZ_REQUIRE(r != nullptr, "roots is NULL");
    LZ_REQUIRE(mode == LZ_TIE_FIRST || mode == LZ_TIE_RANDOM, "unknown tie-break mode");
    r->tiebreak = mode;
    r->seed = seed;
    return LZ_OK;
}

static int players_of(const int32_t *to_play, int n)
{
    int largest = to_play[0];  // cnode.cpp:906-915: players = 1 iff max(virtual_to_play) == -1
    for (int i = 1; i < n; ++i) if (to_play[i] > largest) largest = to_play[i];
    return largest == -1 ? 1 : 2;
}

// (re)seed the random streams of a handle that is being re-armed for a new Roots object with a pinned seed: the device-resident
// epoch (advanced by every prepare) restarts, so that "same seed" means "same streams" whatever the handle did before
extern "C" int lz_roots_reseed(lz_roots *r, uint64_t seed)
{
    LZ_REQUIRE(r != nullptr, "roots is NULL");
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    r->seed = seed;
    if (r->t.rng_epoch) LZ_HIP_CHECK(hipMemsetAsync(r->t.rng_epoch, 0, 4, r->eng->stream));
    return LZ_OK;
}

extern "C" int lz_roots_prepare(lz_roots *r, float root_noise_weight, const float *h_noises_flat,
                                const float *h_value_prefix, const float *h_policy_logits, const int32_t *h_to_play)
{
    LZ_REQUIRE(r != nullptr, "roots is NULL");
    LZ_REQUIRE(h_value_prefix && h_policy_logits && h_to_play, "NULL input");
    const lz_tree_dev &t = r->t;
    const int B = t.B, A = t.A;
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    // staging layout: [noises B*A][noise_off B][vp B][logits B*A][to_play B]
    const size_t need = ((size_t)B * A * 2 + (size_t)B * 3) * 4 + (size_t)B * A * 4 + (size_t)B * 4;
    int rc = ensure_stage(r, need);
    if (rc != LZ_OK) return rc;
    hipStream_t s = r->eng->stream;
    char *h = (char *)r->h_stage, *d = (char *)r->d_stage;
    size_t o_nz = 0, o_off = o_nz + (size_t)B * A * 4, o_vp = o_off + (size_t)B * 4, o_lg = o_vp + (size_t)B * 4,
           o_tp = o_lg + (size_t)B * A * 4, o_nl = o_tp + (size_t)B * 4, total = o_nl + (size_t)B * 4;
    // noise offsets need the legal counts: fetch them (tiny)
    int32_t *hn = (int32_t *)(h + o_nl);
    LZ_HIP_CHECK(hipMemcpyAsync(hn, t.n_legal, (size_t)B * 4, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    int32_t *hoff = (int32_t *)(h + o_off);
    size_t acc = 0;
    for (int i = 0; i < B; ++i) { hoff[i] = (int32_t)acc; acc += hn[i]; }
    if (h_noises_flat) memcpy(h + o_nz, h_noises_flat, acc * 4);
    memcpy(h + o_vp, h_value_prefix, (size_t)B * 4);
    memcpy(h + o_lg, h_policy_logits, (size_t)B * A * 4);
    memcpy(h + o_tp, h_to_play, (size_t)B * 4);
    (void)total;
    LZ_HIP_CHECK(hipMemcpyAsync(d, h, o_nl, hipMemcpyHostToDevice, s));
    lz_tree_launch_prepare(t, root_noise_weight, h_noises_flat ? (const float *)(d + o_nz) : nullptr, 1,
                           (const int32_t *)(d + o_off), (const float *)(d + o_vp), (const float *)(d + o_lg),
                           (const int32_t *)(d + o_tp), s);   // (k_prepare bumps the random-stream epoch)
    LZ_HIP_CHECK(hipGetLastError());
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    r->players = players_of(h_to_play, B);
    r->h_to_play.assign(h_to_play, h_to_play + B);   // lz_roots_adopt_inference uploads it for the fused search
    r->prepared = true;
    r->traverse_count = 0;
    return LZ_OK;
}

extern "C" int lz_roots_prepare_device(lz_roots *r, float root_noise_weight, const float *d_noises,
                                       const float *d_value_prefix, const float *d_policy_logits,
                                       const int32_t *d_to_play, int players)
{
    LZ_REQUIRE(r != nullptr, "roots is NULL");
    LZ_REQUIRE(d_value_prefix && d_policy_logits && d_to_play, "NULL input");
    LZ_REQUIRE(players == 1 || players == 2, "players must be 1 or 2");
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    lz_tree_launch_prepare(r->t, root_noise_weight, d_noises, 0, nullptr, d_value_prefix, d_policy_logits, d_to_play,
                           r->eng->stream);
    LZ_HIP_CHECK(hipGetLastError());
    r->players = players;
    r->prepared = true;
    r->traverse_count = 0;
    return LZ_OK;
}

extern "C" int lz_batch_traverse(lz_roots *r, int pb_c_base, float pb_c_init, float discount_factor,
                                 int32_t *h_virtual_to_play, int32_t *h_out_index_in_search_path,
                                 int32_t *h_out_index_in_batch, int32_t *h_out_last_actions, int32_t *h_out_search_lens)
{
    LZ_REQUIRE(r != nullptr, "roots is NULL");
    LZ_REQUIRE(r->prepared, "batch_traverse before Roots.prepare");
    LZ_REQUIRE(h_virtual_to_play && h_out_index_in_search_path && h_out_index_in_batch && h_out_last_actions && h_out_search_lens, "NULL buffer");
    const lz_tree_dev &t = r->t;
    const int B = t.B;
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    int rc = ensure_stage(r, (size_t)B * 4 * 6);
    if (rc != LZ_OK) return rc;
    hipStream_t s = r->eng->stream;
    memcpy(r->h_stage, h_virtual_to_play, (size_t)B * 4);
    LZ_HIP_CHECK(hipMemcpyAsync(r->d_stage, r->h_stage, (size_t)B * 4, hipMemcpyHostToDevice, s));
    lz_traverse_args a;
    a.pb_c_base = pb_c_base; a.pb_c_init = pb_c_init; a.discount = discount_factor;
    a.players = players_of(h_virtual_to_play, B);
    a.tiebreak = r->tiebreak; a.seed = r->seed; a.counter = r->traverse_count++;
    r->players = a.players;
    lz_tree_launch_traverse(t, a, r->delta, (const int32_t *)r->d_stage, s);
    LZ_HIP_CHECK(hipGetLastError());
    // res_ix .. res_vtp are contiguous [5][B]
    LZ_HIP_CHECK(hipMemcpyAsync(r->h_stage, t.res_ix, (size_t)B * 4 * 5, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    const int32_t *h = (const int32_t *)r->h_stage;
    memcpy(h_out_index_in_search_path, h, (size_t)B * 4);
    memcpy(h_out_index_in_batch, h + B, (size_t)B * 4);
    memcpy(h_out_last_actions, h + 2 * B, (size_t)B * 4);
    memcpy(h_out_search_lens, h + 3 * B, (size_t)B * 4);
    memcpy(h_virtual_to_play, h + 4 * B, (size_t)B * 4);
    return LZ_OK;
}

extern "C" int lz_batch_backpropagate(lz_roots *r, int current_latent_state_index, float discount_factor,
                                      const float *h_value_prefixs, const float *h_values, const float *h_policy_logits,
                                      const int32_t *h_is_reset, const int32_t *h_to_play)
{
    LZ_REQUIRE(r != nullptr, "roots is NULL");
    LZ_REQUIRE(r->prepared, "batch_backpropagate before Roots.prepare");
    LZ_REQUIRE(h_value_prefixs && h_values && h_policy_logits && h_to_play, "NULL input");
    const lz_tree_dev &t = r->t;
    const int B = t.B, A = t.A;
    if (current_latent_state_index < 1 || current_latent_state_index >= t.NN) {
        lz_set_error("current_latent_state_index %d outside the node pool [1,%d]: create the roots with a larger max_simulations",
                     current_latent_state_index, t.NN - 1);
        return LZ_ERR_STATE;
    }
    LZ_REQUIRE(t.variant == LZ_TREE_MUZERO || h_is_reset != nullptr, "is_reset_list is required for the EfficientZero tree");
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    const size_t need = ((size_t)B * 4 + (size_t)B * A) * 4;
    int rc = ensure_stage(r, need);
    if (rc != LZ_OK) return rc;
    hipStream_t s = r->eng->stream;
    char *h = (char *)r->h_stage, *d = (char *)r->d_stage;
    const size_t o_vp = 0, o_v = (size_t)B * 4, o_rst = o_v + (size_t)B * 4, o_tp = o_rst + (size_t)B * 4, o_lg = o_tp + (size_t)B * 4;
    memcpy(h + o_vp, h_value_prefixs, (size_t)B * 4);
    memcpy(h + o_v, h_values, (size_t)B * 4);
    if (h_is_reset) memcpy(h + o_rst, h_is_reset, (size_t)B * 4); else memset(h + o_rst, 0, (size_t)B * 4);
    memcpy(h + o_tp, h_to_play, (size_t)B * 4);
    memcpy(h + o_lg, h_policy_logits, (size_t)B * A * 4);
    LZ_HIP_CHECK(hipMemcpyAsync(d, h, need, hipMemcpyHostToDevice, s));
    lz_tree_launch_backprop(t, current_latent_state_index, discount_factor, (const float *)(d + o_vp), (const float *)(d + o_v),
                            (const float *)(d + o_lg), (const int32_t *)(d + o_rst), 0, (const int32_t *)(d + o_tp), s);
    LZ_HIP_CHECK(hipGetLastError());
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    return LZ_OK;
}

// ---- Gumbel MuZero (gmz_tree.pyx): Roots.prepare / prepare_no_noise, batch_traverse, batch_back_propagate, get_policies,
// get_children_values.  Create the roots with lz_roots_create(..., LZ_TREE_GUMBEL_MUZERO, ...).
static int gumbel_check(lz_roots *r)
{
    LZ_REQUIRE(r != nullptr, "roots is NULL");
    LZ_REQUIRE(r->t.variant == LZ_TREE_GUMBEL_MUZERO, "not a Gumbel MuZero roots handle (lz_roots_create with LZ_TREE_GUMBEL_MUZERO)");
    return LZ_OK;
}

// get_sequence_of_considered_visits (cnode.cpp:1041-1076)
static std::vector<int32_t> considered_visits(int max_num_considered_actions, int num_simulations)
{
    std::vector<int32_t> seq;
    if (max_num_considered_actions <= 1) {
        for (int i = 0; i < num_simulations; ++i) seq.push_back(i);
        return seq;
    }
    const int log2max = (int)std::ceil(std::log2((double)max_num_considered_actions));
    std::vector<int32_t> visits((size_t)max_num_considered_actions, 0);
    int num_considered = max_num_considered_actions;
    while ((int)seq.size() < num_simulations) {
        const int num_extra = std::max(1, num_simulations / (log2max * num_considered));
        for (int i = 0; i < num_extra; ++i) {
            seq.insert(seq.end(), visits.begin(), visits.begin() + num_considered);
            for (int j = 0; j < num_considered; ++j) visits[j] += 1;
        }
        num_considered = std::max(2, num_considered / 2);
    }
    seq.resize((size_t)num_simulations);
    return seq;
}

int lz_groots_set_considered(lz_roots *r, int num_simulations, int max_num_considered_actions, hipStream_t s)
{
    if (r->g_sims == num_simulations && r->g_m == max_num_considered_actions) return LZ_OK;
    if (num_simulations < 1 || num_simulations >= r->t.NN) {
        lz_set_error("num_simulations %d exceeds the node pool (max_simulations %d)", num_simulations, r->t.NN - 1);
        return LZ_ERR_STATE;
    }
    const std::vector<int32_t> seq = considered_visits(std::min(max_num_considered_actions, num_simulations), num_simulations);
    LZ_HIP_CHECK(hipMemcpyAsync(r->t.considered, seq.data(), seq.size() * 4, hipMemcpyHostToDevice, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    r->g_sims = num_simulations;
    r->g_m = max_num_considered_actions;
    return LZ_OK;
}

extern "C" int lz_groots_prepare(lz_roots *r, float root_noise_weight, const float *h_noises_flat, const float *h_rewards,
                                 const float *h_values, const float *h_policy_logits, const int32_t *h_to_play)
{
    int rc = gumbel_check(r);
    if (rc != LZ_OK) return rc;
    LZ_REQUIRE(h_rewards && h_values && h_policy_logits && h_to_play, "NULL input");
    const lz_tree_dev &t = r->t;
    const size_t B = t.B, A = t.A;
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    hipStream_t s = r->eng->stream;
    size_t n_noise = 0;
    std::vector<int32_t> off(B);
    if (h_noises_flat) {
        std::vector<int32_t> nl(B);
        LZ_HIP_CHECK(hipMemcpyAsync(nl.data(), t.n_legal, B * 4, hipMemcpyDeviceToHost, s));
        LZ_HIP_CHECK(hipStreamSynchronize(s));
        for (size_t i = 0; i < B; ++i) { off[i] = (int32_t)n_noise; n_noise += nl[i]; }
    }
    const size_t o_r = 0, o_v = B * 4, o_tp = 2 * B * 4, o_off = 3 * B * 4, o_lg = 4 * B * 4, o_nz = o_lg + B * A * 4, need = o_nz + (n_noise + 1) * 4;
    rc = ensure_stage(r, need);
    if (rc != LZ_OK) return rc;
    char *h = (char *)r->h_stage, *d = (char *)r->d_stage;
    memcpy(h + o_r, h_rewards, B * 4);
    memcpy(h + o_v, h_values, B * 4);
    memcpy(h + o_tp, h_to_play, B * 4);
    memcpy(h + o_off, off.data(), B * 4);
    memcpy(h + o_lg, h_policy_logits, B * A * 4);
    if (h_noises_flat) memcpy(h + o_nz, h_noises_flat, n_noise * 4);
    LZ_HIP_CHECK(hipMemcpyAsync(d, h, need, hipMemcpyHostToDevice, s));
    lz_gtree_launch_prepare(t, root_noise_weight, h_noises_flat ? (const float *)(d + o_nz) : nullptr, 1, (const int32_t *)(d + o_off),
                            (const float *)(d + o_r), (const float *)(d + o_v), (const float *)(d + o_lg), (const int32_t *)(d + o_tp), s);
    LZ_HIP_CHECK(hipGetLastError());
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    r->players = 1;
    r->h_to_play.assign(h_to_play, h_to_play + r->t.B);
    r->prepared = true;
    r->traverse_count = 0;
    return LZ_OK;
}

extern "C" int lz_gbatch_traverse(lz_roots *r, int num_simulations, int max_num_considered_actions, float discount_factor,
                                  int32_t *h_virtual_to_play, int32_t *h_out_index_in_search_path, int32_t *h_out_index_in_batch,
                                  int32_t *h_out_last_actions, int32_t *h_out_search_lens)
{
    int rc = gumbel_check(r);
    if (rc != LZ_OK) return rc;
    LZ_REQUIRE(r->prepared, "batch_traverse before Roots.prepare");
    LZ_REQUIRE(h_virtual_to_play && h_out_index_in_search_path && h_out_index_in_batch && h_out_last_actions && h_out_search_lens, "NULL buffer");
    const lz_tree_dev &t = r->t;
    const size_t B = t.B;
    for (size_t i = 0; i < B; ++i) LZ_REQUIRE(h_virtual_to_play[i] == -1, "the Gumbel MuZero tree is single-player (cnode.cpp:618)");
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    hipStream_t s = r->eng->stream;
    rc = lz_groots_set_considered(r, num_simulations, max_num_considered_actions, s);
    if (rc != LZ_OK) return rc;
    rc = ensure_stage(r, B * 4 * 5);
    if (rc != LZ_OK) return rc;
    lz_gtree_launch_traverse(t, discount_factor, s);
    LZ_HIP_CHECK(hipGetLastError());
    LZ_HIP_CHECK(hipMemcpyAsync(r->h_stage, t.res_ix, B * 4 * 5, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    const int32_t *h = (const int32_t *)r->h_stage;
    memcpy(h_out_index_in_search_path, h, B * 4);
    memcpy(h_out_index_in_batch, h + B, B * 4);
    memcpy(h_out_last_actions, h + 2 * B, B * 4);
    memcpy(h_out_search_lens, h + 3 * B, B * 4);
    return LZ_OK;
}

extern "C" int lz_gbatch_back_propagate(lz_roots *r, int current_latent_state_index, float discount_factor, const float *h_rewards,
                                        const float *h_values, const float *h_policy_logits)
{
    int rc = gumbel_check(r);
    if (rc != LZ_OK) return rc;
    LZ_REQUIRE(r->prepared, "batch_back_propagate before Roots.prepare");
    LZ_REQUIRE(h_rewards && h_values && h_policy_logits, "NULL input");
    const lz_tree_dev &t = r->t;
    const size_t B = t.B, A = t.A;
    if (current_latent_state_index < 1 || current_latent_state_index >= t.NN) {
        lz_set_error("current_latent_state_index %d outside the node pool [1,%d]", current_latent_state_index, t.NN - 1);
        return LZ_ERR_STATE;
    }
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    const size_t o_r = 0, o_v = B * 4, o_lg = 2 * B * 4, need = o_lg + B * A * 4;
    rc = ensure_stage(r, need);
    if (rc != LZ_OK) return rc;
    hipStream_t s = r->eng->stream;
    char *h = (char *)r->h_stage, *d = (char *)r->d_stage;
    memcpy(h + o_r, h_rewards, B * 4);
    memcpy(h + o_v, h_values, B * 4);
    memcpy(h + o_lg, h_policy_logits, B * A * 4);
    LZ_HIP_CHECK(hipMemcpyAsync(d, h, need, hipMemcpyHostToDevice, s));
    lz_gtree_launch_backprop(t, current_latent_state_index, discount_factor, (const float *)(d + o_r), (const float *)(d + o_v),
                             (const float *)(d + o_lg), s);
    LZ_HIP_CHECK(hipGetLastError());
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    return LZ_OK;
}

// Roots.get_policies / get_children_values: [root_num][action_space_size] each; either output may be NULL
extern "C" int lz_groots_get_policies(lz_roots *r, float discount_factor, float *h_out_policies, float *h_out_children_values)
{
    int rc = gumbel_check(r);
    if (rc != LZ_OK) return rc;
    LZ_REQUIRE(r->prepared && (h_out_policies || h_out_children_values), "roots not prepared / no output");
    const lz_tree_dev &t = r->t;
    const size_t B = t.B, A = t.A;
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    rc = ensure_stage(r, 2 * B * A * 4);
    if (rc != LZ_OK) return rc;
    hipStream_t s = r->eng->stream;
    float *dp = (float *)r->d_stage, *dv = dp + B * A;
    lz_gtree_launch_policies(t, discount_factor, h_out_policies ? dp : nullptr, h_out_children_values ? dv : nullptr, s);
    LZ_HIP_CHECK(hipGetLastError());
    LZ_HIP_CHECK(hipMemcpyAsync(r->h_stage, r->d_stage, 2 * B * A * 4, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    if (h_out_policies) memcpy(h_out_policies, r->h_stage, B * A * 4);
    if (h_out_children_values) memcpy(h_out_children_values, (float *)r->h_stage + B * A, B * A * 4);
    return LZ_OK;
}

// ---- ReZero: batch_traverse_with_reuse / batch_backpropagate_with_reuse (ez_tree.pyx:94-121, mz_tree.pyx:84-110)
extern "C" int lz_batch_traverse_with_reuse(lz_roots *r, int pb_c_base, float pb_c_init, float discount_factor,
                                            int32_t *h_virtual_to_play, const int32_t *h_true_action, const float *h_reuse_value,
                                            int32_t *h_out_index_in_search_path, int32_t *h_out_index_in_batch,
                                            int32_t *h_out_last_actions, int32_t *h_out_search_lens)
{
    LZ_REQUIRE(r != nullptr, "roots is NULL");
    LZ_REQUIRE(r->prepared, "batch_traverse_with_reuse before Roots.prepare");
    LZ_REQUIRE(r->t.variant != LZ_TREE_SAMPLED_EFFICIENTZERO, "the sampled tree has no reuse variant");
    LZ_REQUIRE(h_virtual_to_play && h_true_action && h_reuse_value && h_out_index_in_search_path && h_out_index_in_batch &&
               h_out_last_actions && h_out_search_lens, "NULL buffer");
    const lz_tree_dev &t = r->t;
    const size_t B = t.B;
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    int rc = ensure_stage(r, B * 4 * 6);
    if (rc != LZ_OK) return rc;
    hipStream_t s = r->eng->stream;
    char *h = (char *)r->h_stage, *d = (char *)r->d_stage;
    memcpy(h, h_virtual_to_play, B * 4);
    memcpy(h + B * 4, h_true_action, B * 4);
    memcpy(h + B * 8, h_reuse_value, B * 4);
    LZ_HIP_CHECK(hipMemcpyAsync(d, h, B * 12, hipMemcpyHostToDevice, s));
    lz_traverse_args a;
    a.pb_c_base = pb_c_base; a.pb_c_init = pb_c_init; a.discount = discount_factor;
    a.players = players_of(h_virtual_to_play, (int)B);
    a.tiebreak = r->tiebreak; a.seed = r->seed; a.counter = r->traverse_count++;
    r->players = a.players;
    lz_tree_launch_traverse_reuse(t, a, r->delta, (const int32_t *)d, (const int32_t *)(d + B * 4), (const float *)(d + B * 8), s);
    LZ_HIP_CHECK(hipGetLastError());
    LZ_HIP_CHECK(hipMemcpyAsync(h, t.res_ix, B * 4 * 5, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipMemcpyAsync(h + B * 20, t.res_noinf, B * 4, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    const int32_t *hi = (const int32_t *)h;
    for (size_t i = 0; i < B; ++i) h_out_index_in_search_path[i] = hi[5 * B + i] ? -1 : hi[i];  // cnode.cpp:1049
    memcpy(h_out_index_in_batch, hi + B, B * 4);
    memcpy(h_out_last_actions, hi + 2 * B, B * 4);
    memcpy(h_out_search_lens, hi + 3 * B, B * 4);
    memcpy(h_virtual_to_play, hi + 4 * B, B * 4);
    return LZ_OK;
}

extern "C" int lz_batch_backpropagate_with_reuse(lz_roots *r, int current_latent_state_index, float discount_factor,
                                                 const float *h_value_prefixs, const float *h_values, const float *h_policy_logits,
                                                 int n_infer, const int32_t *h_is_reset, const int32_t *h_to_play,
                                                 const int32_t *h_no_inference_lst, const int32_t *h_reuse_lst,
                                                 const float *h_reuse_value)
{
    LZ_REQUIRE(r != nullptr, "roots is NULL");
    LZ_REQUIRE(r->prepared, "batch_backpropagate_with_reuse before Roots.prepare");
    LZ_REQUIRE(h_to_play && h_no_inference_lst && h_reuse_lst && h_reuse_value && n_infer >= 0, "NULL input");
    LZ_REQUIRE(n_infer == 0 || (h_value_prefixs && h_values && h_policy_logits), "network outputs missing");
    const lz_tree_dev &t = r->t;
    const size_t B = t.B, A = t.A;
    if (current_latent_state_index < 1 || current_latent_state_index >= t.NN) {
        lz_set_error("current_latent_state_index %d outside the node pool [1,%d]", current_latent_state_index, t.NN - 1);
        return LZ_ERR_STATE;
    }
    LZ_REQUIRE(t.variant == LZ_TREE_MUZERO || h_is_reset != nullptr, "is_reset_list is required for the EfficientZero tree");
    // the reference walks the two ascending, -1 terminated lists with running counters (cnode.cpp:622-641)
    std::vector<int32_t> mode(B), row(B);
    size_t ca = 0, cb = 0, cc = 0;
    for (size_t i = 0; i < B; ++i) {
        if ((int32_t)i == h_no_inference_lst[ca]) { ++ca; mode[i] = 1; row[i] = 0; }
        else {
            mode[i] = 0;
            if ((int32_t)i == h_reuse_lst[cc]) { mode[i] = 2; ++cc; }
            row[i] = (int32_t)cb++;
        }
    }
    if ((int)cb != n_infer) { lz_set_error("%zu roots need network outputs but %d rows were passed", cb, n_infer); return LZ_ERR_INVALID; }
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    const size_t n = (size_t)n_infer;
    const size_t o_vp = 0, o_v = o_vp + (n + 1) * 4, o_lg = o_v + (n + 1) * 4, o_rst = o_lg + (n + 1) * A * 4, o_tp = o_rst + B * 4,
                 o_md = o_tp + B * 4, o_row = o_md + B * 4, o_rv = o_row + B * 4, need = o_rv + B * 4;
    int rc = ensure_stage(r, need);
    if (rc != LZ_OK) return rc;
    hipStream_t s = r->eng->stream;
    char *h = (char *)r->h_stage, *d = (char *)r->d_stage;
    if (n) { memcpy(h + o_vp, h_value_prefixs, n * 4); memcpy(h + o_v, h_values, n * 4); memcpy(h + o_lg, h_policy_logits, n * A * 4); }
    if (h_is_reset) memcpy(h + o_rst, h_is_reset, B * 4); else memset(h + o_rst, 0, B * 4);
    memcpy(h + o_tp, h_to_play, B * 4);
    memcpy(h + o_md, mode.data(), B * 4);
    memcpy(h + o_row, row.data(), B * 4);
    memcpy(h + o_rv, h_reuse_value, B * 4);
    LZ_HIP_CHECK(hipMemcpyAsync(d, h, need, hipMemcpyHostToDevice, s));
    lz_tree_launch_backprop_reuse(t, current_latent_state_index, discount_factor, (const float *)(d + o_vp), (const float *)(d + o_v),
                                  (const float *)(d + o_lg), (const int32_t *)(d + o_rst), 0, (const int32_t *)(d + o_tp),
                                  (const int32_t *)(d + o_md), (const int32_t *)(d + o_row), (const float *)(d + o_rv), nullptr, nullptr, s);
    LZ_HIP_CHECK(hipGetLastError());
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    return LZ_OK;
}

extern "C" int lz_roots_get_distributions(lz_roots *r, int32_t *h_out_dist, int32_t *h_out_count)
{
    LZ_REQUIRE(r != nullptr && h_out_dist != nullptr, "NULL argument");
    const lz_tree_dev &t = r->t;
    const int B = t.B, A = t.A;
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    if (!r->prepared) {  // get_children_distribution of an unexpanded root is empty (cnode.cpp:272)
        for (size_t i = 0; i < (size_t)B * A; ++i) h_out_dist[i] = -1;
        if (h_out_count) memset(h_out_count, 0, (size_t)B * 4);
        return LZ_OK;
    }
    int rc = ensure_stage(r, ((size_t)B * A + 2 * (size_t)B) * 4);
    if (rc != LZ_OK) return rc;
    hipStream_t s = r->eng->stream;
    int32_t *d = (int32_t *)r->d_stage;
    lz_tree_launch_readout(t, d, d + (size_t)B * A, nullptr, s);
    LZ_HIP_CHECK(hipGetLastError());
    LZ_HIP_CHECK(hipMemcpyAsync(r->h_stage, d, ((size_t)B * A + B) * 4, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    memcpy(h_out_dist, r->h_stage, (size_t)B * A * 4);
    if (h_out_count) memcpy(h_out_count, (int32_t *)r->h_stage + (size_t)B * A, (size_t)B * 4);
    return LZ_OK;
}

extern "C" int lz_roots_get_values(lz_roots *r, float *h_out_values)
{
    LZ_REQUIRE(r != nullptr && h_out_values != nullptr, "NULL argument");
    const lz_tree_dev &t = r->t;
    const int B = t.B, A = t.A;
    if (!r->prepared) { memset(h_out_values, 0, (size_t)B * 4); return LZ_OK; }
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    int rc = ensure_stage(r, ((size_t)B * A + 2 * (size_t)B) * 4);
    if (rc != LZ_OK) return rc;
    hipStream_t s = r->eng->stream;
    int32_t *d = (int32_t *)r->d_stage;
    float *dv = (float *)(d + (size_t)B * A + B);
    if (t.variant == LZ_TREE_SAMPLED_EFFICIENTZERO) lz_stree_launch_readout(t, d, dv, s);  // no legal-action lists there
    else lz_tree_launch_readout(t, d, nullptr, dv, s);
    LZ_HIP_CHECK(hipGetLastError());
    LZ_HIP_CHECK(hipMemcpyAsync(r->h_stage, dv, (size_t)B * 4, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    memcpy(h_out_values, r->h_stage, (size_t)B * 4);
    return LZ_OK;
}

extern "C" int lz_roots_get_trajectories(lz_roots *r, int32_t *h_out, int stride)
{
    LZ_REQUIRE(r != nullptr && h_out != nullptr && stride >= 1, "bad argument");
    const lz_tree_dev &t = r->t;
    const int B = t.B;
    if (!r->prepared) { for (int i = 0; i < B; ++i) h_out[(size_t)i * stride] = -1; return LZ_OK; }
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    int rc = ensure_stage(r, (size_t)B * stride * 4);
    if (rc != LZ_OK) return rc;
    hipStream_t s = r->eng->stream;
    lz_tree_launch_trajectories(t, (int32_t *)r->d_stage, stride, s);
    LZ_HIP_CHECK(hipGetLastError());
    LZ_HIP_CHECK(hipMemcpyAsync(r->h_stage, r->d_stage, (size_t)B * stride * 4, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    memcpy(h_out, r->h_stage, (size_t)B * stride * 4);
    return LZ_OK;
}

// observability: the priors of the root's edges after prepare (CNode::prior of the root's children, cnode.cpp:139-150 and, with
// noise, :163-170), [root_num][A] by ACTION (0 for illegal actions)
extern "C" int lz_roots_get_root_priors(lz_roots *r, float *h_out)
{
    LZ_REQUIRE(r != nullptr && h_out != nullptr && r->prepared, "NULL output / roots not prepared");
    const lz_tree_dev &t = r->t;
    LZ_REQUIRE(t.variant != LZ_TREE_SAMPLED_EFFICIENTZERO, "not for sampled roots");
    const size_t B = t.B, A = t.A, NN = t.NN;
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    hipStream_t s = r->eng->stream;
    std::vector<float4> tmp(B * A);
    LZ_HIP_CHECK(hipMemcpy2DAsync(tmp.data(), A * sizeof(float4), t.edge, NN * A * sizeof(float4), A * sizeof(float4), B, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    for (size_t i = 0; i < B * A; ++i) h_out[i] = tmp[i].x;
    return LZ_OK;
}

extern "C" int lz_roots_get_minmax(lz_roots *r, float *h_out)
{
    LZ_REQUIRE(r != nullptr && h_out != nullptr, "NULL argument");
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    hipStream_t s = r->eng->stream;
    int rc = ensure_stage(r, (size_t)r->t.B * 8);
    if (rc != LZ_OK) return rc;
    LZ_HIP_CHECK(hipMemcpyAsync(r->h_stage, r->t.minmax, (size_t)r->t.B * 8, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    memcpy(h_out, r->h_stage, (size_t)r->t.B * 8);
    return LZ_OK;
}

// ------------------------------------------------------------------------------------------------
// select_action on the device (lzero/policy/utils.py:637-661): p_i = N_i^(1/T) / sum, arg-max or one draw, entropy in bits.
// float64 like the Python original; one thread per root (A is small).
// ------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ uint64_t sel_mix64(uint64_t z)
{
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__global__ void k_select_action(lz_tree_dev t, double inv_temperature, int deterministic, uint64_t seed,
                                int32_t *__restrict__ pos_out, double *__restrict__ ent_out)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= t.B) return;
    const bool sampled = t.variant == LZ_TREE_SAMPLED_EFFICIENTZERO;
    const int A = t.A, n = sampled ? A : t.n_legal[b];
    const float4 *edge0 = t.edge + (size_t)b * t.NN * A;
    auto count = [&](int j) -> int {
        const int slot = sampled ? t.rep[(size_t)b * t.NN * A + j] : t.legal[(size_t)b * A + j];
        return __float_as_int(edge0[slot].y);
    };
    // N^(1/T); T = 1 needs no pow (x^1 is x) -- k_collect_rows (lz_search.hip) computes the same expression
    auto powt = [&](int c) -> double { return inv_temperature == 1.0 ? (double)c : pow((double)c, inv_temperature); };
    double sum = 0.0;
    int best = -1, arg = 0;
    for (int j = 0; j < n; ++j) {
        const int c = count(j);
        sum += powt(c);
        if (c > best) { best = c; arg = j; }  // np.argmax: first maximum
    }
    const double u = (double)(sel_mix64(sel_mix64(seed) ^ (uint64_t)b) >> 11) * (1.0 / 9007199254740992.0);  // [0, 1)
    double acc = 0.0, H = 0.0;
    int pick = -1, last = 0;
    for (int j = 0; j < n; ++j) {
        const double p = powt(count(j)) / sum;
        if (p > 0.0) { H -= p * log2(p); last = j; }
        acc += p;
        if (pick < 0 && u < acc) pick = j;  // searchsorted(cumsum(p), u, side='right') like np.random.choice
    }
    if (pick < 0) pick = last;
    pos_out[b] = deterministic ? arg : pick;
    ent_out[b] = H;
}
}  // namespace

void lz_launch_select_action(const lz_tree_dev &t, double inv_temperature, int deterministic, uint64_t seed, int32_t *d_pos,
                             double *d_ent, hipStream_t s)
{
    hipLaunchKernelGGL(k_select_action, dim3((unsigned)((t.B + 63) / 64)), dim3(64), 0, s, t, inv_temperature, deterministic, seed, d_pos, d_ent);
}

extern "C" int lz_roots_select_action(lz_roots *r, double temperature, int deterministic, uint64_t seed, int32_t *h_action_pos,
                                      double *h_entropy)
{
    LZ_REQUIRE(r != nullptr && h_action_pos != nullptr && h_entropy != nullptr, "NULL argument");
    LZ_REQUIRE(r->prepared, "select_action before Roots.prepare");
    LZ_REQUIRE(temperature > 0.0, "temperature must be positive");
    const lz_tree_dev &t = r->t;
    const size_t B = t.B;
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    int rc = ensure_stage(r, B * 16);
    if (rc != LZ_OK) return rc;
    hipStream_t s = r->eng->stream;
    double *d_ent = (double *)r->d_stage;
    int32_t *d_pos = (int32_t *)((char *)r->d_stage + B * 8);
    lz_launch_select_action(t, 1.0 / temperature, deterministic, seed, d_pos, d_ent, s);
    LZ_HIP_CHECK(hipGetLastError());
    LZ_HIP_CHECK(hipMemcpyAsync(r->h_stage, r->d_stage, B * 12, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    memcpy(h_entropy, r->h_stage, B * 8);
    memcpy(h_action_pos, (char *)r->h_stage + B * 8, B * 4);
    return LZ_OK;
}

// ------------------------------------------------------------------------------------------------
// Sampled EfficientZero trees (continuous actions)
// ------------------------------------------------------------------------------------------------
extern "C" int lz_sroots_create(lz_engine *e, int root_num, int action_dim, int num_of_sampled_actions, int max_simulations,
                                lz_roots **out)
{
    LZ_REQUIRE(e != nullptr && out != nullptr, "engine/out is NULL");
    *out = nullptr;
    LZ_REQUIRE(root_num > 0 && max_simulations > 0, "root_num and max_simulations must be positive");
    LZ_REQUIRE(action_dim >= 1 && action_dim <= 64, "action_dim must be in [1, 64]");
    LZ_REQUIRE(num_of_sampled_actions >= 1 && num_of_sampled_actions <= 64, "num_of_sampled_actions must be in [1, 64] (one lane per sampled action)");
    LZ_HIP_CHECK(hipSetDevice(e->device));
    lz_roots *r = nullptr;
    int rc = lz_roots_alloc(e, LZ_TREE_SAMPLED_EFFICIENTZERO, root_num, num_of_sampled_actions, max_simulations, &r, action_dim);
    if (rc != LZ_OK) return rc;
    lz_tree_launch_minmax_reset(r->t, e->stream);
    *out = r;
    return LZ_OK;
}

// discrete action space (continuous_action_space = False): K of the action_space_size actions per node, policy = logits
extern "C" int lz_sroots_create_discrete(lz_engine *e, int root_num, int action_space_size, int num_of_sampled_actions,
                                         int max_simulations, lz_roots **out)
{
    LZ_REQUIRE(action_space_size >= 1 && action_space_size <= 256, "action_space_size must be in [1, 256] (four actions per lane in the draw without replacement)");
    LZ_REQUIRE(num_of_sampled_actions <= action_space_size, "num_of_sampled_actions must not exceed action_space_size (sampling is without replacement)");
    int rc = lz_sroots_create(e, root_num, 1, num_of_sampled_actions, max_simulations, out);
    if (rc != LZ_OK) return rc;
    (*out)->t.disc_A = action_space_size;
    return LZ_OK;
}

static int sampled_check(lz_roots *r)
{
    LZ_REQUIRE(r != nullptr, "roots is NULL");
    LZ_REQUIRE(r->t.variant == LZ_TREE_SAMPLED_EFFICIENTZERO, "not a Sampled-EfficientZero roots handle (use lz_sroots_create)");
    return LZ_OK;
}

extern "C" int lz_sroots_prepare(lz_roots *r, float root_noise_weight, const float *h_noises, const float *h_value_prefix,
                                 const float *h_policy, const int32_t *h_to_play, const float *h_given)
{
    int rc = sampled_check(r);
    if (rc != LZ_OK) return rc;
    LZ_REQUIRE(h_value_prefix && h_policy && h_to_play, "NULL input");
    (void)root_noise_weight; (void)h_noises;  // only perturb priors that the shipped uniform-prior score never reads
    const lz_tree_dev &t = r->t;
    const size_t B = t.B, K = t.A, D = t.D;
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    const size_t PSZ = t.disc_A > 0 ? (size_t)t.disc_A : 2 * D;  // floats per root in the policy array
    const size_t o_vp = 0, o_pol = o_vp + B * 4, o_tp = o_pol + B * PSZ * 4, o_giv = o_tp + B * 4, need = o_giv + B * K * D * 4;
    rc = ensure_stage(r, need);
    if (rc != LZ_OK) return rc;
    char *h = (char *)r->h_stage, *d = (char *)r->d_stage;
    memcpy(h + o_vp, h_value_prefix, B * 4);
    memcpy(h + o_pol, h_policy, B * PSZ * 4);
    memcpy(h + o_tp, h_to_play, B * 4);
    if (h_given) memcpy(h + o_giv, h_given, B * K * D * 4);
    hipStream_t s = r->eng->stream;
    LZ_HIP_CHECK(hipMemcpyAsync(d, h, h_given ? need : o_giv, hipMemcpyHostToDevice, s));
    lz_sample_args sa;
    sa.given = h_given ? (const float *)(d + o_giv) : nullptr;
    sa.policy = (const float *)(d + o_pol);
    sa.seed = r->seed;
    sa.counter = 0;
    lz_stree_launch_prepare(t, sa, (const float *)(d + o_vp), (const int32_t *)(d + o_tp), s);
    lz_tree_launch_bump_epoch(t, s);
    LZ_HIP_CHECK(hipGetLastError());
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    r->players = players_of(h_to_play, (int)B);
    r->h_to_play.assign(h_to_play, h_to_play + B);
    r->prepared = true;
    r->traverse_count = 0;
    return LZ_OK;
}

extern "C" int lz_sbatch_traverse(lz_roots *r, int pb_c_base, float pb_c_init, float discount_factor, int32_t *h_virtual_to_play,
                                  int32_t *h_out_index_in_search_path, int32_t *h_out_index_in_batch, float *h_out_last_actions,
                                  int32_t *h_out_search_lens)
{
    int rc = sampled_check(r);
    if (rc != LZ_OK) return rc;
    LZ_REQUIRE(r->prepared, "batch_traverse before Roots.prepare");
    LZ_REQUIRE(h_virtual_to_play && h_out_index_in_search_path && h_out_index_in_batch && h_out_last_actions && h_out_search_lens, "NULL buffer");
    const lz_tree_dev &t = r->t;
    const size_t B = t.B, D = t.D;
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    rc = ensure_stage(r, B * 4 * 6 + B * D * 4);
    if (rc != LZ_OK) return rc;
    hipStream_t s = r->eng->stream;
    memcpy(r->h_stage, h_virtual_to_play, B * 4);
    LZ_HIP_CHECK(hipMemcpyAsync(r->d_stage, r->h_stage, B * 4, hipMemcpyHostToDevice, s));
    lz_traverse_args a;
    a.pb_c_base = pb_c_base; a.pb_c_init = pb_c_init; a.discount = discount_factor;
    a.players = players_of(h_virtual_to_play, (int)B);
    a.tiebreak = r->tiebreak; a.seed = r->seed; a.counter = r->traverse_count++;
    r->players = a.players;
    lz_stree_launch_traverse(t, a, r->delta, (const int32_t *)r->d_stage, s);
    LZ_HIP_CHECK(hipGetLastError());
    LZ_HIP_CHECK(hipMemcpyAsync(r->h_stage, t.res_ix, B * 4 * 5, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipMemcpyAsync((char *)r->h_stage + B * 4 * 5, t.res_last_action_f, B * D * 4, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    const int32_t *h = (const int32_t *)r->h_stage;
    memcpy(h_out_index_in_search_path, h, B * 4);
    memcpy(h_out_index_in_batch, h + B, B * 4);
    memcpy(h_out_search_lens, h + 3 * B, B * 4);
    memcpy(h_virtual_to_play, h + 4 * B, B * 4);
    memcpy(h_out_last_actions, (char *)r->h_stage + B * 4 * 5, B * D * 4);
    return LZ_OK;
}

extern "C" int lz_sbatch_backpropagate(lz_roots *r, int current_latent_state_index, float discount_factor,
                                       const float *h_value_prefixs, const float *h_values, const float *h_policy,
                                       const int32_t *h_is_reset, const int32_t *h_to_play, const float *h_given)
{
    int rc = sampled_check(r);
    if (rc != LZ_OK) return rc;
    LZ_REQUIRE(r->prepared, "batch_backpropagate before Roots.prepare");
    LZ_REQUIRE(h_value_prefixs && h_values && h_policy && h_is_reset && h_to_play, "NULL input");
    const lz_tree_dev &t = r->t;
    const size_t B = t.B, K = t.A, D = t.D;
    if (current_latent_state_index < 1 || current_latent_state_index >= t.NN) {
        lz_set_error("current_latent_state_index %d outside the node pool [1,%d]", current_latent_state_index, t.NN - 1);
        return LZ_ERR_STATE;
    }
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    const size_t PSZ = t.disc_A > 0 ? (size_t)t.disc_A : 2 * D;
    const size_t o_vp = 0, o_v = B * 4, o_rst = 2 * B * 4, o_tp = 3 * B * 4, o_pol = 4 * B * 4, o_giv = o_pol + B * PSZ * 4,
                 need = o_giv + B * K * D * 4;
    rc = ensure_stage(r, need);
    if (rc != LZ_OK) return rc;
    char *h = (char *)r->h_stage, *d = (char *)r->d_stage;
    memcpy(h + o_vp, h_value_prefixs, B * 4);
    memcpy(h + o_v, h_values, B * 4);
    memcpy(h + o_rst, h_is_reset, B * 4);
    memcpy(h + o_tp, h_to_play, B * 4);
    memcpy(h + o_pol, h_policy, B * PSZ * 4);
    if (h_given) memcpy(h + o_giv, h_given, B * K * D * 4);
    hipStream_t s = r->eng->stream;
    LZ_HIP_CHECK(hipMemcpyAsync(d, h, h_given ? need : o_giv, hipMemcpyHostToDevice, s));
    lz_sample_args sa;
    sa.given = h_given ? (const float *)(d + o_giv) : nullptr;
    sa.policy = (const float *)(d + o_pol);
    sa.seed = r->seed;
    sa.counter = (uint32_t)current_latent_state_index;
    lz_stree_launch_backprop(t, current_latent_state_index, discount_factor, (const float *)(d + o_vp), (const float *)(d + o_v), sa,
                             (const int32_t *)(d + o_rst), 0, (const int32_t *)(d + o_tp), s);
    LZ_HIP_CHECK(hipGetLastError());
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    return LZ_OK;
}

extern "C" int lz_sroots_get_distributions(lz_roots *r, int32_t *h_out)
{
    int rc = sampled_check(r);
    if (rc != LZ_OK) return rc;
    LZ_REQUIRE(h_out != nullptr && r->prepared, "NULL output / roots not prepared");
    const lz_tree_dev &t = r->t;
    const size_t B = t.B, K = t.A;
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    rc = ensure_stage(r, B * K * 4);
    if (rc != LZ_OK) return rc;
    hipStream_t s = r->eng->stream;
    lz_stree_launch_readout(t, (int32_t *)r->d_stage, nullptr, s);
    LZ_HIP_CHECK(hipGetLastError());
    LZ_HIP_CHECK(hipMemcpyAsync(r->h_stage, r->d_stage, B * K * 4, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    memcpy(h_out, r->h_stage, B * K * 4);
    return LZ_OK;
}

extern "C" int lz_sroots_get_sampled_actions(lz_roots *r, float *h_out)
{
    int rc = sampled_check(r);
    if (rc != LZ_OK) return rc;
    LZ_REQUIRE(h_out != nullptr && r->prepared, "NULL output / roots not prepared");
    const lz_tree_dev &t = r->t;
    const size_t B = t.B, K = t.A, D = t.D, NN = t.NN;
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    hipStream_t s = r->eng->stream;
    // actions of node 0 of every root: [B] strided blocks of K*D floats
    LZ_HIP_CHECK(hipMemcpy2DAsync(h_out, K * D * 4, t.actions, NN * K * D * 4, K * D * 4, B, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    return LZ_OK;
}

// the K actions of expanded node `node` (0 = root, s + 1 = the node expanded by simulation s) of every root: what the
// reference keeps in CNode::legal_actions after expand (cnode.cpp:238-327).  Observability for the exact replay gate of the
// fused sampled search: the device's own draws are injected into the CPU oracle.
extern "C" int lz_sroots_get_node_actions(lz_roots *r, int node, float *h_out)
{
    int rc = sampled_check(r);
    if (rc != LZ_OK) return rc;
    LZ_REQUIRE(h_out != nullptr && r->prepared, "NULL output / roots not prepared");
    const lz_tree_dev &t = r->t;
    LZ_REQUIRE(node >= 0 && node < t.NN, "node out of range");
    const size_t B = t.B, K = t.A, D = t.D, NN = t.NN;
    LZ_HIP_CHECK(hipSetDevice(r->eng->device));
    hipStream_t s = r->eng->stream;
    LZ_HIP_CHECK(hipMemcpy2DAsync(h_out, K * D * 4, t.actions + (size_t)node * K * D, NN * K * D * 4, K * D * 4, B, hipMemcpyDeviceToHost, s));
    LZ_HIP_CHECK(hipStreamSynchronize(s));
    return LZ_OK;
}
