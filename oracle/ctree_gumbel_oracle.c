/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement (plain C11, scalar) of the reference's Gumbel MuZero tree
 *   lzero/mcts/ctree/ctree_gumbel_muzero/lib/cnode.cpp  (LightZero v0.2.0)
 * Nothing on the product path may include, link or call this file.
 *
 * The tree is deterministic: every node's Gumbel vector is drawn from std::mt19937(0) (gumbel_rng = 0.0 is never changed,
 * cnode.cpp:58-59,86-89,1133-1151), selection is an arg-max that keeps the first maximum (:724-733, :771-781).
 * libstdc++'s std::extreme_value_distribution<float> over std::mt19937 is restated below (generate_canonical<float, 24>).
 *
 * Parity pin: tests/test_gumbel_oracle.py drives this file and the compiled reference (oracle/_ref/stock/gmz_tree*.so)
 * with the same inputs and requires identical per-simulation records, visit counts, root values, improved policies and
 * completed values; tests/golden/gumbel_*.npz holds vectors generated from the compiled reference.
 * Compile with -ffp-contract=off (the reference is baseline x86-64, no FMA).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define G_FLOAT_MAX 1000000.0f
#define G_FLOAT_MIN (-G_FLOAT_MAX)

typedef struct {
    int visit_count, to_play, latent_index, batch_index, best_action;
    float reward, raw_value, prior, value_sum;
    int expanded, first_child, n_legal, legal_off; /* legal_off < 0: 0..A-1 */
} GNode;

typedef struct { float maximum, minimum, value_delta_max; } GMinMax;

typedef struct {
    int B, A, cap;
    GNode *nodes;
    int *n_nodes, *legal, *n_legal, *path, *path_len;
    GMinMax *mm;
    float *gumbel; /* [A]: gumbel_scale * d(gen), the same prefix for every node */
} GTree;

/* ---- std::mt19937 + std::extreme_value_distribution<float>(0, 1), libstdc++ ---- */
typedef struct { uint32_t mt[624]; int idx; } MT;
static void mt_seed(MT *m, uint32_t s)
{
    m->mt[0] = s;
    for (int i = 1; i < 624; ++i) m->mt[i] = 1812433253u * (m->mt[i - 1] ^ (m->mt[i - 1] >> 30)) + (uint32_t)i;
    m->idx = 624;
}
static uint32_t mt_next(MT *m)
{
    if (m->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (m->mt[i] & 0x80000000u) | (m->mt[(i + 1) % 624] & 0x7fffffffu);
            m->mt[i] = m->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        m->idx = 0;
    }
    uint32_t y = m->mt[m->idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}
void gtree_generate_gumbel(float gumbel_scale, float gumbel_rng, int shape, float *out) /* cnode.cpp:1133-1151 */
{
    MT m;
    mt_seed(&m, (uint32_t)gumbel_rng);
    for (int i = 0; i < shape; ++i) {
        /* generate_canonical<float, 24>(mt19937): one 32-bit draw; sum = float(draw) ; ret = sum / 2^32 (float ops) */
        float sum = (float)mt_next(&m);
        float ret = sum / 4294967296.0f;
        if (ret >= 1.0f) ret = nextafterf(1.0f, 0.0f);
        out[i] = gumbel_scale * (0.0f - 1.0f * logf(-logf(1.0f - ret)));
    }
}

static void gnode_init(GNode *n, float prior)
{
    memset(n, 0, sizeof(*n));
    n->prior = prior; n->best_action = -1; n->latent_index = -1; n->batch_index = -1; n->first_child = -1; n->legal_off = -1;
}

GTree *gtree_create(int B, int A, int max_sims, const int *legal_flat, const int *legal_cnt)
{
    GTree *t = (GTree *)calloc(1, sizeof(GTree));
    t->B = B; t->A = A; t->cap = 1 + (max_sims + 1) * A;
    t->nodes = (GNode *)malloc(sizeof(GNode) * (size_t)B * t->cap);
    t->n_nodes = (int *)calloc(B, sizeof(int));
    t->legal = (int *)calloc((size_t)B * A, sizeof(int));
    t->n_legal = (int *)calloc(B, sizeof(int));
    t->path = (int *)calloc((size_t)B * t->cap, sizeof(int));
    t->path_len = (int *)calloc(B, sizeof(int));
    t->mm = (GMinMax *)malloc(sizeof(GMinMax) * B);
    t->gumbel = (float *)malloc(sizeof(float) * A);
    gtree_generate_gumbel(10.0f, 0.0f, A, t->gumbel);
    int off = 0;
    for (int i = 0; i < B; ++i) {
        t->n_legal[i] = legal_cnt[i];
        for (int j = 0; j < legal_cnt[i]; ++j) t->legal[(size_t)i * A + j] = legal_flat[off + j];
        off += legal_cnt[i];
        GNode *root = &t->nodes[(size_t)i * t->cap];
        gnode_init(root, 0.0f);
        root->n_legal = legal_cnt[i]; root->legal_off = 0;
        t->n_nodes[i] = 1;
        t->mm[i].maximum = G_FLOAT_MIN; t->mm[i].minimum = G_FLOAT_MAX; t->mm[i].value_delta_max = 0.0f;
    }
    return t;
}
void gtree_destroy(GTree *t)
{
    if (!t) return;
    free(t->nodes); free(t->n_nodes); free(t->legal); free(t->n_legal); free(t->path); free(t->path_len); free(t->mm); free(t->gumbel); free(t);
}
void gtree_set_delta(GTree *t, float d) { for (int i = 0; i < t->B; ++i) t->mm[i].value_delta_max = d; }

static int legal_at(const GTree *t, int env, const GNode *n, int j) { return n->legal_off < 0 ? j : t->legal[(size_t)env * t->A + j]; }
static float gvalue(const GNode *n) { return n->visit_count == 0 ? 0.0f : n->value_sum / n->visit_count; } /* :245-261 */
static void mm_update(GMinMax *m, float v) { if (v > m->maximum) m->maximum = v; if (v < m->minimum) m->minimum = v; }

/* CNode::expand :94-156 */
static void gexpand(GTree *t, int env, int ni, int to_play, int latent_index, int batch_index, float reward, float value, const float *logits)
{
    GNode *pool = &t->nodes[(size_t)env * t->cap], *n = &pool[ni];
    const int A = t->A;
    n->to_play = to_play; n->latent_index = latent_index; n->batch_index = batch_index; n->reward = reward; n->raw_value = value;
    if (n->n_legal == 0) { n->n_legal = A; n->legal_off = -1; }
    float policy[A];
    float policy_sum = 0.0f, policy_max = G_FLOAT_MIN;
    for (int j = 0; j < n->n_legal; ++j) { int a = legal_at(t, env, n, j); if (policy_max < logits[a]) policy_max = logits[a]; }
    for (int j = 0; j < n->n_legal; ++j) {
        int a = legal_at(t, env, n, j);
        float tp = expf(logits[a] - policy_max);
        policy_sum += tp;
        policy[a] = tp;
    }
    n->first_child = t->n_nodes[env];
    t->n_nodes[env] += A;
    for (int a = 0; a < A; ++a) gnode_init(&pool[n->first_child + a], 0.0f);
    for (int j = 0; j < n->n_legal; ++j) { int a = legal_at(t, env, n, j); pool[n->first_child + a].prior = policy[a] / policy_sum; }
    n->expanded = 1;
}

/* CRoots::prepare / prepare_no_noise :418-455 */
void gtree_prepare(GTree *t, float noise_w, const float *noises_flat, const float *rewards, const float *values, const float *logits, const int *to_play)
{
    int off = 0;
    for (int i = 0; i < t->B; ++i) {
        GNode *pool = &t->nodes[(size_t)i * t->cap], *root = &pool[0];
        gexpand(t, i, 0, to_play[i], 0, i, rewards[i], values[i], logits + (size_t)i * t->A);
        if (noises_flat) {
            for (int j = 0; j < root->n_legal; ++j) {
                GNode *c = &pool[root->first_child + legal_at(t, i, root, j)];
                float prior = c->prior;
                c->prior = prior * (1 - noise_w) + noises_flat[off + j] * noise_w;
            }
            off += root->n_legal;
        }
        root->visit_count += 1;
    }
}

/* csoftmax :903-928 -- `log(sum)` on a float picks the float overload there (checked against the compiled reference:
 * the double variant disagrees on ~30 % of random inputs) */
#ifndef GSOFTMAX_LOG
#define GSOFTMAX_LOG(s) logf(s)
#endif
static void gsoftmax(float *x, int n)
{
    float m = x[0];
    for (int i = 1; i < n; ++i) if (x[i] > m) m = x[i];
    float sum = 0;
    for (int i = 0; i < n; ++i) sum += expf(x[i] - m);
    for (int i = 0; i < n; ++i) x[i] = expf(x[i] - m - GSOFTMAX_LOG(sum));
}

/* qtransform_completed_by_mix_value :984-1037 with the header defaults (cnode.h:101-102): maxvisit_init 50, value_scale 0.1,
 * rescale_values true, epsilon 1e-8.  visit[] / prior[] are per legal position. */
static void completed_q(const GTree *t, int env, const GNode *node, const int *visit, const float *prior, int n, float discount, float *out)
{
    const GNode *pool = &t->nodes[(size_t)env * t->cap];
    float q[n], ptmp[n];
    for (int j = 0; j < n; ++j) {  /* CNode::get_q :181-197 */
        const GNode *c = &pool[node->first_child + legal_at(t, env, node, j)];
        q[j] = c->reward + discount * gvalue(c);
        ptmp[j] = prior[j];
    }
    gsoftmax(ptmp, n);
    /* compute_mixed_value :930-966 */
    float visit_count_sum = 0.0f, probs_sum = 0.0f, weighted_q_sum = 0.0f;
    const float min_num = -10e7f;
    for (int j = 0; j < n; ++j) visit_count_sum += visit[j];
    for (int j = 0; j < n; ++j) ptmp[j] = ptmp[j] > min_num ? ptmp[j] : min_num;
    for (int j = 0; j < n; ++j) if (visit[j] > 0) probs_sum += ptmp[j];
    for (int j = 0; j < n; ++j) if (visit[j] > 0) weighted_q_sum += ptmp[j] * q[j] / probs_sum;
    float value = (node->raw_value + visit_count_sum * weighted_q_sum) / (visit_count_sum + 1);
    for (int j = 0; j < n; ++j) out[j] = visit[j] > 0 ? q[j] : value;
    /* rescale_qvalues :968-982 */
    float mx = out[0], mn = out[0];
    for (int j = 1; j < n; ++j) { if (out[j] > mx) mx = out[j]; if (out[j] < mn) mn = out[j]; }
    float gap = mx - mn;
    gap = gap > 1e-8f ? gap : 1e-8f;
    for (int j = 0; j < n; ++j) out[j] = (out[j] - mn) / gap;
    float max_visit = (float)visit[0];
    for (int j = 1; j < n; ++j) if ((float)visit[j] > max_visit) max_visit = (float)visit[j];
    float visit_scale = 50.0f + max_visit;
    for (int j = 0; j < n; ++j) out[j] = out[j] * visit_scale * 0.1f;
}

/* get_sequence_of_considered_visits :1041-1076 */
void gtree_considered_visits(int max_num_considered_actions, int num_simulations, int *out)
{
    if (max_num_considered_actions <= 1) { for (int i = 0; i < num_simulations; ++i) out[i] = i; return; }
    int log2max = (int)ceil(log2((double)max_num_considered_actions));
    int visits[max_num_considered_actions];
    for (int i = 0; i < max_num_considered_actions; ++i) visits[i] = 0;
    int num_considered = max_num_considered_actions, len = 0;
    int cap = num_simulations + max_num_considered_actions * (num_simulations + 2);
    int *seq = (int *)malloc(sizeof(int) * (size_t)cap);
    while (len < num_simulations) {
        int num_extra = num_simulations / (log2max * num_considered);
        if (num_extra < 1) num_extra = 1;
        for (int i = 0; i < num_extra; ++i) {
            for (int j = 0; j < num_considered && len < cap; ++j) seq[len++] = visits[j];
            for (int j = 0; j < num_considered; ++j) visits[j] += 1;
        }
        num_considered = num_considered / 2 > 2 ? num_considered / 2 : 2;
    }
    for (int i = 0; i < num_simulations; ++i) out[i] = seq[i];
    free(seq);
}

static void child_stats(const GTree *t, int env, const GNode *node, int *visit, float *prior)
{
    const GNode *pool = &t->nodes[(size_t)env * t->cap];
    for (int j = 0; j < node->n_legal; ++j) {
        const GNode *c = &pool[node->first_child + legal_at(t, env, node, j)];
        visit[j] = c->visit_count; prior[j] = c->prior;
    }
}

/* cselect_root_child :701-745 (+ score_considered :1096-1131) */
static int select_root(const GTree *t, int env, const GNode *root, float discount, int num_simulations, int max_considered)
{
    const int n = root->n_legal;
    int visit[n]; float prior[n], cq[n];
    child_stats(t, env, root, visit, prior);
    completed_q(t, env, root, visit, prior, n, discount, cq);
    int num_considered = max_considered < num_simulations ? max_considered : num_simulations;
    int seq[num_simulations];
    gtree_considered_visits(num_considered, num_simulations, seq);
    int simulation_index = 0;
    for (int j = 0; j < n; ++j) simulation_index += visit[j];
    int considered_visit = seq[simulation_index];
    const float low_logit = -1e9f;
    float max_logit = prior[0];
    for (int j = 1; j < n; ++j) if (prior[j] > max_logit) max_logit = prior[j];
    float argmax = -INFINITY;
    int max_action = legal_at(t, env, root, 0);
    for (int j = 0; j < n; ++j) {
        float logit = prior[j] - max_logit;
        float penalty = (visit[j] == considered_visit) ? 0.0f : -INFINITY;
        float s = t->gumbel[j] + logit + cq[j];
        float score = (low_logit > s ? low_logit : s) + penalty;
        if (score > argmax) { argmax = score; max_action = legal_at(t, env, root, j); }
    }
    return max_action;
}

/* cselect_interior_child :747-790 */
static int select_interior(const GTree *t, int env, const GNode *node, float discount)
{
    const int n = node->n_legal;
    int visit[n]; float prior[n], cq[n], probs[n];
    child_stats(t, env, node, visit, prior);
    completed_q(t, env, node, visit, prior, n, discount, cq);
    for (int j = 0; j < n; ++j) probs[j] = prior[j] + cq[j];
    gsoftmax(probs, n);
    int visit_count_sum = 0;
    for (int j = 0; j < n; ++j) visit_count_sum += visit[j];
    float argmax = -INFINITY;
    int max_action = legal_at(t, env, node, 0);
    for (int j = 0; j < n; ++j) {
        float v = probs[j] - (float)visit[j] / (float)(1 + visit_count_sum);
        if (v > argmax) { argmax = v; max_action = legal_at(t, env, node, j); }
    }
    return max_action;
}

/* cbatch_traverse :834-897 */
void gtree_traverse(GTree *t, int num_simulations, int max_considered, float discount, const int *virtual_to_play, int *out_ix, int *out_iy,
                    int *out_last_action, int *out_search_len)
{
    (void)virtual_to_play;
    int last_action = -1;
    for (int i = 0; i < t->B; ++i) {
        GNode *pool = &t->nodes[(size_t)i * t->cap];
        int *path = &t->path[(size_t)i * t->cap];
        int ni = 0, is_root = 1, search_len = 0, plen = 0;
        path[plen++] = ni;
        while (pool[ni].expanded) {
            GNode *node = &pool[ni];
            int action = is_root ? select_root(t, i, node, discount, num_simulations, max_considered) : select_interior(t, i, node, discount);
            is_root = 0;
            node->best_action = action;
            ni = node->first_child + action;
            last_action = action;
            path[plen++] = ni;
            search_len += 1;
        }
        const GNode *parent = &pool[path[plen - 2]];
        out_ix[i] = parent->latent_index; out_iy[i] = parent->batch_index;
        out_last_action[i] = last_action; out_search_len[i] = search_len;
        t->path_len[i] = plen;
    }
}

/* cbatch_back_propagate :633-652 + cback_propagate :605-631 */
void gtree_backpropagate(GTree *t, int latent_index, float discount, const float *rewards, const float *values, const float *logits, const int *to_play)
{
    for (int i = 0; i < t->B; ++i) {
        GNode *pool = &t->nodes[(size_t)i * t->cap];
        const int *path = &t->path[(size_t)i * t->cap];
        int leaf = path[t->path_len[i] - 1];
        gexpand(t, i, leaf, to_play[i], latent_index, i, rewards[i], values[i], logits + (size_t)i * t->A);
        float bootstrap_value = values[i];
        for (int k = t->path_len[i] - 1; k >= 0; --k) {
            GNode *node = &pool[path[k]];
            node->value_sum += bootstrap_value;
            node->visit_count += 1;
            float true_reward = node->reward;
            mm_update(&t->mm[i], true_reward + discount * gvalue(node));
            bootstrap_value = true_reward + discount * bootstrap_value;
        }
    }
}

void gtree_get_distributions(const GTree *t, int *out, int *out_cnt)
{
    for (int i = 0; i < t->B; ++i) {
        const GNode *pool = &t->nodes[(size_t)i * t->cap], *root = &pool[0];
        int n = root->expanded ? root->n_legal : 0;
        out_cnt[i] = n;
        for (int j = 0; j < t->A; ++j) out[(size_t)i * t->A + j] = -1;
        for (int j = 0; j < n; ++j) out[(size_t)i * t->A + j] = pool[root->first_child + legal_at(t, i, root, j)].visit_count;
    }
}
void gtree_get_values(const GTree *t, float *out) { for (int i = 0; i < t->B; ++i) out[i] = gvalue(&t->nodes[(size_t)i * t->cap]); }

/* CNode::get_children_value :309-338 / CNode::get_policy :350-375, [B][A] */
void gtree_get_children_values(const GTree *t, float discount, float *out)
{
    for (int i = 0; i < t->B; ++i) {
        const GNode *root = &t->nodes[(size_t)i * t->cap];
        const int n = root->n_legal;
        int visit[n]; float prior[n], cq[n];
        child_stats(t, i, root, visit, prior);
        completed_q(t, i, root, visit, prior, n, discount, cq);
        for (int a = 0; a < t->A; ++a) out[(size_t)i * t->A + a] = -INFINITY;
        for (int j = 0; j < n; ++j) out[(size_t)i * t->A + legal_at(t, i, root, j)] = cq[j];
    }
}
void gtree_get_policies(const GTree *t, float discount, float *out)
{
    for (int i = 0; i < t->B; ++i) {
        const GNode *root = &t->nodes[(size_t)i * t->cap];
        const int n = root->n_legal;
        int visit[n]; float prior[n], cq[n];
        child_stats(t, i, root, visit, prior);
        completed_q(t, i, root, visit, prior, n, discount, cq);
        float *p = out + (size_t)i * t->A;
        for (int a = 0; a < t->A; ++a) p[a] = -INFINITY;
        for (int j = 0; j < n; ++j) p[legal_at(t, i, root, j)] = prior[j] + cq[j];
        gsoftmax(p, t->A);
    }
}
void gtree_get_trajectories(const GTree *t, int *out, int stride)
{
    for (int i = 0; i < t->B; ++i) {
        const GNode *pool = &t->nodes[(size_t)i * t->cap], *node = &pool[0];
        int k = 0, best = node->best_action;
        while (best >= 0 && k < stride - 1) { out[(size_t)i * stride + k++] = best; node = &pool[node->first_child + best]; best = node->best_action; }
        out[(size_t)i * stride + k] = -1;
    }
}
void gtree_softmax(float *x, int n) { gsoftmax(x, n); }
