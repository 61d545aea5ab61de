// Internal (non-ABI) definitions shared by the translation units of liblz_mi355.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/lz_mi355.h"
#include "lz_nn_kernels.h"

void lz_set_error(const char *fmt, ...);

#define LZ_HIP_CHECK(expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            lz_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return LZ_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

#define LZ_REQUIRE(cond, msg)                    \
    do {                                         \
        if (!(cond)) {                           \
            lz_set_error("%s (%s)", msg, #cond); \
            return LZ_ERR_INVALID;               \
        }                                        \
    } while (0)

// Every device allocation of the library goes through here.  LZ_POISON=<byte> (debug) fills fresh allocations with
// that byte so that a kernel depending on what the allocator handed back shows up as a parity failure
// (tests/test_poison_gpu.py runs the search under 0x00, 0x7f and 0xff and requires identical results).
#include <stdlib.h>
static inline hipError_t lz_dev_malloc(void **p, size_t bytes)
{
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return e;
    const char *v = getenv("LZ_POISON");
    if (v && *v) {
        e = hipMemset(*p, (int)strtol(v, nullptr, 0) & 0xff, bytes);
        if (e == hipSuccess) e = hipDeviceSynchronize();  // the engine stream does not order against the null stream
    }
    return e;
}

struct lz_model;  // network weights + workspaces (lz_nn.hip)

struct lz_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    lz_model *model = nullptr;
    uint64_t model_uid = 0;      // bumped by lz_model_create: a different network (shapes) now lives on this engine
    uint64_t weights_gen = 0;    // bumped whenever a device weight pointer changed (lz_model_finalize that had to re-allocate)
    // optional in-stream timing of one kernel class (bench.py roofline): HIP event pairs recorded on the
    // engine stream around every launch of the tagged kernel while enabled
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;   // 2 * capacity events, created on enable
    size_t prof_used = 0;              // pairs recorded
};

// HBM-resident node arrays of a batch of trees.
//
// Expanded node n of root b  <=>  latent-state index n (root = 0; the node expanded by simulation s
// is s + 1 == `current_latent_state_index`, mcts_ctree.py:870).  A node's statistics live on the
// EDGE that leads to it (the parent's child slot), so that one pUCT selection step reads A
// consecutive 16-byte records: edge[b][n][a] = {prior, visit_count, value_sum, value_prefix(child)}.
struct lz_tree_dev {
    int B, A, NN;               // roots, action space, nodes per root (= max_simulations + 1)
    int variant;
    float4 *edge;               // [B][NN][A]  {prior, visit(int bits), value_sum, child value_prefix / reward}
    int32_t *child;             // [B][NN][A]  expanded-node index of the child, -1 = not expanded, -2 = illegal
    float *node_vp;             // [B][NN]     value_prefix / reward of the expanded node itself
    int32_t *node_reset;        // [B][NN]     is_reset (EZ)
    int32_t *node_to_play;      // [B][NN]
    int32_t *node_best;         // [B][NN]     best_action (last selected action at this node)
    int32_t *root_visit;        // [B]
    float *root_vsum;           // [B]
    int32_t *legal;             // [B][A]      root legal-action list, order preserved
    int32_t *n_legal;           // [B]         (A and identity list when the reference list is empty)
    float *minmax;              // [B][2]      (minimum, maximum)
    // results of the last traverse (CSearchResults, cnode.h:66-80)
    int32_t *path_node;         // [B][NN]     expanded nodes on the search path, root first
    int32_t *path_act;          // [B][NN]     action taken at path_node[k]
    int32_t *res_ix, *res_iy, *res_last_action, *res_search_len, *res_vtp;  // [B] each
    uint32_t *rng_epoch;        // [1] incremented by every prepare (stochastic tie-break stream)
    // Gumbel MuZero (variant 3)
    float *node_raw;            // [B][NN]     CNode::raw_value: the value head's output at the node
    float *gsoft;               // [B][NN][A]  Gumbel tree: csoftmax of a node's child priors by legal position, computed once when the
                                //             node is expanded (root: after the noise) instead of at every visit
    float *gumbel;              // [A]         gumbel_scale * extreme_value(mt19937(0)): the same prefix for every node (cnode.cpp:86-89)
    int32_t *considered;        // [NN]        get_sequence_of_considered_visits(min(m, S), S) of the current search
    int32_t *node_bidx;         // [B][NN]     CNode::batch_index (== b except under ReZero's packed inference batches)
    uint64_t *node_link;        // [B][NN]     parent node | action at the parent | depth of every expanded node (lz_link_pack; root 0):
                                //             lets every node of a tree be scored at once (dev_traverse_par)
    int32_t *res_noinf;         // [B]         ReZero: the last traverse ended on an already expanded node (reference index -1)
    // Sampled EfficientZero (variant 2, continuous actions): A == K sampled actions per node
    int D;                      // action dimension
    int disc_A;                 // > 0: discrete action space of that size (D == 1, an action is the float of its index; policy = logits)
    int32_t *rep;               // [B][NN][K]  position of the first legal action with the same "%f" key (shared child)
    int32_t *nchild;            // [B][NN]     number of distinct keys == children.size()
    float *actions;             // [B][NN][K][D] sampled actions (legal_actions of every expanded node)
    float *res_last_action_f;   // [B][D]      last selected action of the latest traverse
};

struct lz_graph_key {
    int sims, pb_c_base;
    float pb_c_init, discount;
    int horizon;
    float delta;
    int players, tiebreak;
    uint64_t seed;
    uint64_t knobs;   // the debugging switches that change the captured launch sequence (so that toggling one re-captures)
    uint64_t model_uid, weights_gen;  // the kernel arguments hold weight pointers: a re-created model or re-allocated weights re-capture
    int trace;        // tracing adds one device-to-device copy per simulation to the captured sequence (bit 1: head debug buffers)
    int stamps;       // in-graph timing stamps change kernel arguments
};

struct lz_roots {
    lz_engine *eng = nullptr;
    lz_tree_dev t{};
    void *slab = nullptr;        // one allocation backing every array above
    size_t slab_bytes = 0;
    float delta = 0.0f;
    int tiebreak = LZ_TIE_FIRST;
    uint64_t seed = 0;
    uint32_t traverse_count = 0;
    int players = 1;             // from the last traverse's virtual_to_play (cnode.cpp:906-915)
    bool prepared = false;
    // pinned host staging for the fine-grained (host-pointer) API
    void *h_stage = nullptr;
    void *d_stage = nullptr;
    size_t stage_bytes = 0;
    hipEvent_t stage_done = nullptr;   // recorded behind an upload from h_stage that was left in flight (lz_roots_upload_legal)
    bool stage_pending = false;
    // ---- fused search state (allocated on first lz_initial_inference), all in HBM
    void *pool_slab = nullptr;
    uint64_t pool_model_uid = 0;    // the model (lz_engine::model_uid) whose shapes sized pool_slab / d_obs / d_results
    size_t d_obs_bytes = 0, results_bytes = 0;
    float *latent_pool = nullptr;   // [NN][B][HW][C]   NHWC latent of every expanded node
    float *h_pool = nullptr;        // [NN][B][H]       LSTM state pools (EfficientZero)
    float *c_pool = nullptr;
    float *explore_tab = nullptr;   // [128] lz_traverse_args::tab of the running search (inside the slab)
    int tab_base = -1; float tab_init = 0.0f; bool tab_valid = false;   // what explore_tab was last filled for (lz_search refills it on a change: no launch per search)
    float *sim_vp = nullptr;        // [NN][B]          value prefix / reward of node n (after h^-1)
    float *sim_value = nullptr;     // [NN][B]
    float *sim_logits = nullptr;    // [NN][B][A]
    float *t_x1 = nullptr, *t_x2 = nullptr, *t_x3 = nullptr;  // [B][HW][C] scratch activations
    float *t_rx = nullptr;          // [B][HW*HC] reward conv output
    float *t_pv = nullptr;          // [B][HW][2*HC] value | policy conv outputs
    bool trace_on = false;          // record res_* of every simulation (parity tests)
    bool head_debug = false;        // lz_roots_enable_trace(on & 2): every simulation's support-wide head logits and pre-transform
                                    // expectations go to hd_logits / hd_expect (split heads included)
    float *hd_logits = nullptr;     // [NN][2: value, value prefix | reward][B][SUP]   (own allocation, only while head_debug)
    float *hd_expect = nullptr;     // [NN][2][B]                                       (same allocation)
    size_t hd_sup = 0;
    // in-graph timing (bench.py roofline; lz_roots_enable_stamps): constant-rate (100 MHz) s_memrealtime stamps written by the
    // first / last workgroup of the two launches of a simulation -- [NN][4] = {start, end} of the chain launch, then of the LSTM launch
    unsigned long long *stamps = nullptr;
    bool stamps_on = false;
    float *t_hbn = nullptr;         // [B][H]
    float *dbg_logits[2] = {nullptr, nullptr};  // [B][support]
    int32_t *trace = nullptr;       // [NN][5][B]  copies of res_* per simulation
    int32_t *d_to_play = nullptr;   // [B]
    float *d_zero_vp = nullptr;     // [B] zeros
    float *d_noise = nullptr;       // [B][A]
    int32_t *d_noise_off = nullptr; // [B]
    float *d_obs = nullptr;         // staging for lz_initial_inference_host
    const float *last_obs = nullptr; // the observation batch of the latest lz_initial_inference (caller's or d_obs): the newest
                                    // frame of every env-step row comes from here (lz_roots_collect_rows)
    float *sh_part = nullptr;       // split heads: [B][3 heads][H/16 unit tiles][32] first-layer partial sums of the head MLPs (LSTM launch)
    void *fuse_ctl = nullptr;       // LZ_SIM_ONE_LAUNCH=1 (k_sim_fused): lz_res_ctl + one arrival counter per (launch, 16-root group); zeroed once per search
    size_t fuse_ctl_bytes = 0;
    bool fuse_used = false;         // the last enqueued search ran fused launches: lz_search checks lz_res_ctl::fault behind it
    float *mt[14] = {};             // MLP model family: [B][Wmax] scratch activations (lz_mlp.hip)
    std::vector<int32_t> h_n_legal;  // host copy of n_legal (noise offsets without a device round trip)
    std::vector<int32_t> h_to_play;  // to_play of the last HOST-side prepare (lz_roots_prepare & co.): lz_roots_adopt_inference
    void *h_prep = nullptr;          // pinned staging of prepare_from_inference (noise | offsets | to_play), own buffer so that the
    size_t prep_bytes = 0;           // upload can stay asynchronous
    hipEvent_t prep_done = nullptr;  // recorded after that upload: the buffer is rewritten only once it has fired
    void *d_results = nullptr, *h_results = nullptr;  // lz_roots_get_search_results: one packed block, device + pinned host
    // env-step rows in flight (lz_roots_collect_rows_begin / _end): the event behind the row kernel and where the header words land
    hipEvent_t rows_done = nullptr;
    bool rows_pending = false, rows_logits = false;
    float *rows_hh = nullptr;
    size_t rows_B = 0, rows_hw = 0, rows_pa = 0;
    int g_sims = -1, g_m = -1;      // Gumbel MuZero: (num_simulations, max_num_considered_actions) of the uploaded visit table
    void *d_reuse = nullptr;        // ReZero fused search: true_action [B] | reuse_value [B] | per-simulation inference counts [NN]
    float *d_given = nullptr;       // Sampled-EZ parity runs: [records][B][K][D] injected draws (record 0 = roots, s + 1 = simulation s)
    int given_records = 0;
    hipGraphExec_t graph_exec = nullptr;  // captured search (lz_search)
    lz_graph_key graph_key{};
    bool inferred = false;
    bool inference_fresh = false;   // lz_initial_inference ran and no prepare has used it yet
};

// lz_tree.hip launchers (all asynchronous on `stream`)
struct lz_traverse_args {
    int pb_c_base;
    float pb_c_init, discount;
    int players;
    int tiebreak;
    uint64_t seed;
    uint32_t counter;
    int fresh_minmax = 0;      // k_traverse only: the first selection of a search starts a fresh CMinMaxStats (cminimax.cpp:6-10) itself
    int serial = 0;            // 1 (LZ_TRAVERSE_SERIAL): the LDS tree step walks level by level (dev_traverse) even where the
                               // tree-parallel selection (dev_traverse_par) applies -- A/B runs and the bit-identity test
    const float *tab = nullptr;  // [2][64] exploration factors by visit count n (lz_tree_launch_explore_tab): log((n + base + 1) / base) + init | sqrt(n);
                               // null: dev_step_lds computes them itself (a software logf + two table fetches in front of every step)
    unsigned long long *dbg_ts = nullptr;  // timing experiments (debug build, LZ_DEBUG_TREE_SEP_TS): [64 roots][8] cycle stamps of k_backprop_traverse
};
// one expand + backup + next-selection step for every root (dev_step_lds in lz_tree_dev.h), as run by k_backprop_traverse_lds
// or by the convolution chain's prologue (lz_launch_chain with a step)
struct lz_tree_step {
    lz_tree_dev t;
    int new_node;              // = latent slot of the leaf being expanded
    float discount;
    const float *vps, *values, *logits;  // the leaf's network outputs [B], [B], [B][A]
    int horizon;
    lz_traverse_args a;
    float delta;
    const int32_t *vtp;
    unsigned long long *ts;    // timing experiments (debug build, LZ_DEBUG_TREE_TS): s_memtime stamps of root 0's step; null in production
    lz_split_heads sh;         // sh.on: the leaf's network outputs are not in vps / values / logits yet -- the other waves of the chain
                               // launch compute them from the LSTM launch's partials and hand them over in LDS (k_chain_w)
};
// select_action for every root (lz_capi.hip): d_pos [B] int32, d_ent [B] float64
void lz_launch_select_action(const lz_tree_dev &t, double inv_temperature, int deterministic, uint64_t seed, int32_t *d_pos,
                             double *d_ent, hipStream_t s);
void lz_tree_launch_minmax_reset(const lz_tree_dev &t, hipStream_t s);
void lz_tree_launch_explore_tab(float *out, int pb_c_base, float pb_c_init, hipStream_t s);   // out[0..63], out[64..127]: see lz_traverse_args::tab
void lz_tree_launch_prepare(const lz_tree_dev &t, float noise_w, const float *d_noises, int noises_ragged,
                            const int32_t *d_noise_off, const float *d_vp, const float *d_logits,
                            const int32_t *d_to_play, hipStream_t s);
void lz_tree_launch_traverse(const lz_tree_dev &t, const lz_traverse_args &a, float delta, const int32_t *d_vtp_in,
                             hipStream_t s);
// d_is_reset may be null; when horizon > 0 the kernel derives is_reset = (search_len % horizon == 0)
// (mcts_ctree.py:859) itself and also writes it to d_is_reset_out if that is non-null.
void lz_tree_launch_backprop(const lz_tree_dev &t, int latent_index, float discount, const float *d_vp,
                             const float *d_values, const float *d_logits, const int32_t *d_is_reset, int horizon,
                             const int32_t *d_to_play, hipStream_t s);
void lz_tree_launch_backprop_traverse(const lz_tree_dev &t, int latent_index, float discount, const float *d_vp,
                                      const float *d_values, const float *d_logits, int horizon,
                                      const lz_traverse_args &a, float delta, const int32_t *d_vtp_in, hipStream_t s);
void lz_tree_launch_bump_epoch(const lz_tree_dev &t, hipStream_t s);
// lz_tree_wide.hip: the same four steps for action spaces beyond 256 (a node's children walked in 64-lane chunks by a loop instead of
// living in at most four register chunks); the launchers above dispatch here
void lz_tree_wide_launch_prepare(const lz_tree_dev &t, float noise_w, const float *d_noises, int noises_ragged, const int32_t *d_noise_off,
                                 const float *d_vp, const float *d_logits, const int32_t *d_to_play, hipStream_t s);
void lz_tree_wide_launch_traverse(const lz_tree_dev &t, const lz_traverse_args &a, float delta, const int32_t *d_vtp_in, hipStream_t s);
void lz_tree_wide_launch_backprop(const lz_tree_dev &t, int latent_index, float discount, const float *d_vp, const float *d_values,
                                  const float *d_logits, const int32_t *d_is_reset, int horizon, const int32_t *d_to_play, hipStream_t s);
void lz_tree_wide_launch_traverse_reuse(const lz_tree_dev &t, const lz_traverse_args &a, float delta, const int32_t *d_vtp_in,
                                        const int32_t *d_true_action, const float *d_reuse_value, hipStream_t s);
void lz_tree_wide_launch_backprop_reuse(const lz_tree_dev &t, int latent_index, float discount, const float *d_vp, const float *d_values,
                                        const float *d_logits, const int32_t *d_is_reset, int horizon, const int32_t *d_to_play,
                                        const int32_t *d_mode, const int32_t *d_row, const float *d_reuse_value,
                                        const int32_t *d_true_action, int32_t *d_infer_counter, hipStream_t s);
void lz_tree_wide_launch_backprop_traverse(const lz_tree_dev &t, int latent_index, float discount, const float *d_vp, const float *d_values,
                                           const float *d_logits, int horizon, const lz_traverse_args &a, float delta,
                                           const int32_t *d_vtp_in, hipStream_t s);
// Gumbel MuZero (ctree_gumbel_muzero/lib/cnode.cpp): selection by sequential halving at the root and by the completed-Q
// improved policy below it; expand / backup are the MuZero kernels (+ raw value), readout adds get_policies / get_children_values
int lz_groots_set_considered(lz_roots *r, int num_simulations, int max_num_considered_actions, hipStream_t s);
void lz_gtree_launch_prepare(const lz_tree_dev &t, float noise_w, const float *d_noises, int noises_ragged, const int32_t *d_noise_off,
                             const float *d_rewards, const float *d_values, const float *d_logits, const int32_t *d_to_play, hipStream_t s);
void lz_gtree_launch_traverse(const lz_tree_dev &t, float discount, hipStream_t s);
void lz_gtree_launch_backprop(const lz_tree_dev &t, int latent_index, float discount, const float *d_rewards, const float *d_values,
                              const float *d_logits, hipStream_t s);
void lz_gtree_launch_backprop_traverse(const lz_tree_dev &t, int latent_index, float discount, const float *d_rewards,
                                       const float *d_values, const float *d_logits, hipStream_t s);
void lz_gtree_launch_policies(const lz_tree_dev &t, float discount, float *d_policies, float *d_children_values, hipStream_t s);
// ReZero search_with_reuse (cnode.cpp:603-649, 697-754, 816-884, 965-1072)
void lz_tree_launch_traverse_reuse(const lz_tree_dev &t, const lz_traverse_args &a, float delta, const int32_t *d_vtp_in,
                                   const int32_t *d_true_action, const float *d_reuse_value, hipStream_t s);
void lz_tree_launch_backprop_reuse(const lz_tree_dev &t, int latent_index, float discount, const float *d_vp, const float *d_values,
                                   const float *d_logits, const int32_t *d_is_reset, int horizon, const int32_t *d_to_play,
                                   const int32_t *d_mode, const int32_t *d_row, const float *d_reuse_value,
                                   const int32_t *d_true_action, int32_t *d_infer_counter, hipStream_t s);
void lz_tree_launch_readout(const lz_tree_dev &t, int32_t *d_dist, int32_t *d_cnt, float *d_values, hipStream_t s);
void lz_tree_launch_trajectories(const lz_tree_dev &t, int32_t *d_out, int stride, hipStream_t s);
// Sampled EfficientZero tree (lz_tree_sampled.hip)
struct lz_sample_args {
    const float *given;     // optional [B][K][D] already-sampled (post-tanh) actions; null => drawn on the device
    const float *policy;    // [B][2D] (mu | sigma)
    uint64_t seed;
    uint32_t counter;
};
void lz_stree_launch_prepare(const lz_tree_dev &t, const lz_sample_args &sa, const float *d_vp, const int32_t *d_to_play, hipStream_t s);
void lz_stree_launch_traverse(const lz_tree_dev &t, const lz_traverse_args &a, float delta, const int32_t *d_vtp_in, hipStream_t s);
void lz_stree_launch_backprop(const lz_tree_dev &t, int latent_index, float discount, const float *d_vp, const float *d_values,
                              const lz_sample_args &sa, const int32_t *d_is_reset, int horizon, const int32_t *d_to_play, hipStream_t s);
void lz_stree_launch_backprop_traverse(const lz_tree_dev &t, int latent_index, float discount, const float *d_vp, const float *d_values,
                                       const lz_sample_args &sa, int horizon, const lz_traverse_args &a, float delta,
                                       const int32_t *d_vtp_in, hipStream_t s);
void lz_stree_launch_readout(const lz_tree_dev &t, int32_t *d_dist, float *d_values, hipStream_t s);
