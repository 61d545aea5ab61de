/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement (plain C11, scalar) of the reference's Sampled-EfficientZero tree: continuous
 * action spaces (K actions ~ tanh(N(mu, sigma))) and discrete ones (K of the A actions drawn without replacement by sorting
 * u_i^(1/p_i), cnode.cpp:290-327).  Nothing on the product path may include, link or call this file.
 *
 * Restates (reference = /root/reference, LightZero v0.2.0):
 *   lzero/mcts/ctree/ctree_sampled_efficientzero/lib/cnode.cpp
 *       CAction / hash keys :55-110, expand (continuous branch) :194-282,:408-452, add_exploration_noise :454-479,
 *       compute_mean_q :480-520, cbackpropagate :860-945, cbatch_backpropagate :947-966, cselect_child :968-1024,
 *       cucb_score :1026-1108, cbatch_traverse :1110-1187, get_distributions / get_sampled_actions / get_values :712-800
 * and the pieces of libstdc++ (GCC 11, the image's compiler) that expand() uses to draw its K actions:
 *   std::default_random_engine == minstd_rand0 (x <- 16807 x mod 2^31-1), seeded per expand with the low 32 bits of
 *   std::chrono::system_clock::now() (cnode.cpp:251); std::generate_canonical<float,24> (one engine draw, divided by
 *   float(2147483646.0L) = 2^31); std::normal_distribution<float> = Marsaglia polar method, a FRESH distribution object
 *   per (sample, dimension) (cnode.cpp:268), so the cached second variate is always discarded.
 *
 * Shipped behaviour reproduced deliberately ("Sampled-EZ"):
 *   * `empirical_distribution_type.compare("density")` is 0 (equal), so the prior term of the UCB score is the
 *     "uniform" branch  pb_c * 1 / parent->children.size()  (cnode.cpp:1054-1079); the log-prob priors and the
 *     Dirichlet noise mixed into them (:470-472) never reach a score.  They are still computed here.
 *   * children are keyed by a hash of std::to_string(float) ("%f", 6 decimals): two sampled actions whose every
 *     dimension prints identically share ONE child node, while both stay in legal_actions (:430-432).
 *
 * Pin: tests/test_sampled_oracle.py drives this file and the reference's own compiled module (oracle/_ref/det,
 * built with rand() -> 0 and system_clock::now() replaced by a settable counter) with the same clock values and
 * requires identical sampled actions, per-simulation (ix, iy, last_action, search_len), visit counts and root values.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define S_FLOAT_MAX 1000000.0f
#define S_FLOAT_MIN (-S_FLOAT_MAX)

typedef struct {
    int visit_count, to_play, latent_index, batch_index, is_reset;
    int best_slot;     /* legal-action position of best_action; -1 = root sentinel (is_root_action == 1) */
    float value_prefix, prior, value_sum;
    int expanded;
    int first_child;   /* K consecutive node slots; legal action i uses slot first_child + rep[i] */
    int n_children;    /* children.size(): distinct hash keys */
} SNode;

typedef struct { float maximum, minimum, value_delta_max; } SMinMax;

typedef struct {
    int B, D, K, cap, tiebreak;
    int A_disc;        /* > 0: discrete action space of that size (D == 1, an action is the float of its index) */
    int pstride;       /* floats per root in the policy arrays: 2 D (mu | sigma) or A_disc (logits) */
    uint64_t clock;    /* what system_clock::now().time_since_epoch().count() returns next */
    SNode *nodes;      /* [B][cap] */
    int *n_nodes;      /* [B] */
    float *actions;    /* [B][cap_exp][K][D]: sampled actions of expanded node e (legal_actions) */
    int *rep;          /* [B][cap_exp][K]: representative position of legal action i (first i' with the same key) */
    int *exp_of_node;  /* [B][cap]: expansion record index of a node, -1 if not expanded */
    int *n_exp;        /* [B] */
    int cap_exp;
    SMinMax *mm;
    int *path, *path_len;
} STree;

/* ---- libstdc++ pieces ---- */
static uint32_t lcg_next(uint64_t *x) { *x = (*x * 16807ull) % 2147483647ull; return (uint32_t)*x; }

static float canonical(uint64_t *x) /* generate_canonical<float, 24>(minstd_rand0): m == 1 */
{
    float sum = (float)(lcg_next(x) - 1u) * 1.0f;
    const float tmp = 2147483648.0f; /* float(1.0f * 2147483646.0L) */
    float ret = sum / tmp;
    if (ret >= 1.0f) ret = nextafterf(1.0f, 0.0f);
    return ret;
}

static float normal_fresh(uint64_t *x, float mean, float stddev) /* normal_distribution<float>{mean,stddev}(gen) */
{
    float vx, vy, r2;
    do {
        vx = (float)(2.0f * canonical(x) - 1.0);
        vy = (float)(2.0f * canonical(x) - 1.0);
        r2 = vx * vx + vy * vy;
    } while (r2 > 1.0 || r2 == 0.0);
    const float mult = sqrtf(-2 * logf(r2) / r2);
    float ret = vy * mult;
    ret = ret * stddev + mean;
    return ret;
}

/* generate_canonical<double, 53>(minstd_rand0): two engine draws (log2 of the range is 30), libstdc++ random.tcc */
static double canonical53(uint64_t *x)
{
    const long double r = 2147483646.0L;  /* max() - min() + 1 */
    double sum = 0, tmp = 1;
    for (int k = 2; k != 0; --k) {
        sum += (double)(lcg_next(x) - 1u) * tmp;
        tmp = (double)((long double)tmp * r);
    }
    double ret = sum / tmp;
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    return ret;
}

/* ---- tree ---- */
STree *stree_create(int B, int D, int K, int max_sims)
{
    STree *t = (STree *)calloc(1, sizeof(STree));
    t->B = B; t->D = D; t->K = K; t->tiebreak = 0; t->clock = 1; t->A_disc = 0; t->pstride = 2 * D;
    t->cap_exp = max_sims + 1;
    t->cap = 1 + t->cap_exp * K;
    t->nodes = (SNode *)calloc((size_t)B * t->cap, sizeof(SNode));
    t->n_nodes = (int *)calloc(B, sizeof(int));
    t->actions = (float *)calloc((size_t)B * t->cap_exp * K * D, sizeof(float));
    t->rep = (int *)calloc((size_t)B * t->cap_exp * K, sizeof(int));
    t->exp_of_node = (int *)malloc(sizeof(int) * (size_t)B * t->cap);
    t->n_exp = (int *)calloc(B, sizeof(int));
    t->mm = (SMinMax *)malloc(sizeof(SMinMax) * B);
    t->path = (int *)calloc((size_t)B * t->cap, sizeof(int));
    t->path_len = (int *)calloc(B, sizeof(int));
    for (size_t i = 0; i < (size_t)B * t->cap; ++i) t->exp_of_node[i] = -1;
    for (int i = 0; i < B; ++i) {
        SNode *r = &t->nodes[(size_t)i * t->cap];
        memset(r, 0, sizeof(*r));
        r->best_slot = -1; r->latent_index = -1; r->batch_index = -1;
        t->n_nodes[i] = 1;
        t->mm[i].maximum = S_FLOAT_MIN; t->mm[i].minimum = S_FLOAT_MAX; t->mm[i].value_delta_max = 0.0f;
    }
    return t;
}

/* discrete action space of size A: D == 1, policy = A logits */
STree *stree_create_discrete(int B, int A, int K, int max_sims)
{
    STree *t = stree_create(B, 1, K, max_sims);
    t->A_disc = A; t->pstride = A;
    return t;
}

void stree_destroy(STree *t)
{
    if (!t) return;
    free(t->nodes); free(t->n_nodes); free(t->actions); free(t->rep); free(t->exp_of_node); free(t->n_exp);
    free(t->mm); free(t->path); free(t->path_len); free(t);
}

void stree_set_clock(STree *t, uint64_t c) { t->clock = c; }
void stree_set_tiebreak(STree *t, int m) { t->tiebreak = m; }
void stree_set_delta(STree *t, float d) { for (int i = 0; i < t->B; ++i) t->mm[i].value_delta_max = d; }

static void mm_update(SMinMax *m, float v) { if (v > m->maximum) m->maximum = v; if (v < m->minimum) m->minimum = v; }
static float mm_normalize(const SMinMax *m, float value)
{
    float norm_value = value, delta = m->maximum - m->minimum;
    if (delta > 0) {
        if (delta < m->value_delta_max) norm_value = (norm_value - m->minimum) / m->value_delta_max;
        else norm_value = (norm_value - m->minimum) / delta;
    }
    return norm_value;
}
static float node_value(const SNode *n) { return n->visit_count == 0 ? 0.0f : n->value_sum / n->visit_count; }

/* CNode::expand, continuous branch (cnode.cpp:194-282, 408-452).  policy = [mu_0..mu_{D-1}, sigma_0..sigma_{D-1}].
 * If `given` is non-NULL it holds K*D already-sampled (post-tanh) actions and no random numbers are drawn. */
static void node_expand(STree *t, int env, int ni, int to_play, int latent_index, int batch_index, float value_prefix,
                        const float *policy, const float *given)
{
    SNode *pool = &t->nodes[(size_t)env * t->cap];
    SNode *n = &pool[ni];
    const int K = t->K, D = t->D;
    n->to_play = to_play; n->latent_index = latent_index; n->batch_index = batch_index; n->value_prefix = value_prefix;
    const int e = t->n_exp[env]++;
    t->exp_of_node[(size_t)env * t->cap + ni] = e;
    float *act = &t->actions[(((size_t)env * t->cap_exp) + e) * K * D];
    int *rep = &t->rep[(((size_t)env * t->cap_exp) + e) * K];
    float logp_after[K];
    uint64_t x = 0;
    if (!given) {
        const uint32_t seed = (uint32_t)(t->clock++); /* unsigned seed = system_clock::now()...count() */
        x = seed % 2147483647ull;
        if (x == 0) x = 1;
    }
    if (t->A_disc) {
        /* discrete branch (cnode.cpp:288-327, 434-447): probs = exp(l) / (sum exp(l) + 1e-6), keys u^(1/p) sorted in
         * descending order, the first K indices are the sampled actions, prior = probs[action] */
        const int A = t->A_disc;
        float probs[A];
        float logits_exp_sum = 0;
        for (int a = 0; a < A; ++a) logits_exp_sum += expf(policy[a]);
        for (int a = 0; a < A; ++a) probs[a] = (float)((double)expf(policy[a]) / ((double)logits_exp_sum + 1e-6));
        if (given) {
            for (int i = 0; i < K; ++i) { act[i] = given[i]; logp_after[i] = probs[(int)given[i]]; }
        } else {
            int idx[A];
            double key[A];
            for (int a = 0; a < A; ++a) { idx[a] = a; key[a] = pow(canonical53(&x), 1. / (double)probs[a]); }
            /* std::sort(begin, end, cmp) with cmp(x, y) = x.second > y.second: for n <= 16 libstdc++ runs its insertion sort
             * (bits/stl_algo.h __insertion_sort); equal keys keep their order there.  Larger n go through introsort, whose
             * order among EQUAL keys is not restated (distinct keys sort identically under any algorithm). */
            for (int i = 1; i < A; ++i) {
                const int vi = idx[i]; const double vk = key[i];
                if (vk > key[0]) { memmove(&idx[1], &idx[0], sizeof(int) * i); memmove(&key[1], &key[0], sizeof(double) * i); idx[0] = vi; key[0] = vk; }
                else { int j = i; while (vk > key[j - 1]) { idx[j] = idx[j - 1]; key[j] = key[j - 1]; --j; } idx[j] = vi; key[j] = vk; }
            }
            for (int i = 0; i < K; ++i) { act[i] = (float)idx[i]; logp_after[i] = probs[idx[i]]; }
        }
    }
    for (int i = 0; i < K && !t->A_disc; ++i) {
        float prob_before = 1;
        double ysum = 0.;
        for (int j = 0; j < D; ++j) {
            const float mu = policy[j], sigma = policy[D + j];
            float s;
            if (given) {
                act[i * D + j] = given[i * D + j];
                continue;
            }
            s = normal_fresh(&x, mu, sigma);
            prob_before = (float)((double)prob_before *
                                  exp(-pow((double)(s - mu), 2) / (2 * pow((double)sigma, 2)) - (double)logf(sigma) -
                                      log(sqrt(2 * M_PI))));
            act[i * D + j] = tanhf(s);
            const float yj = (float)(1 - pow((double)tanhf(s), 2) + 1e-6);
            ysum += (double)yj; /* std::accumulate(y.begin(), y.end(), 0.) */
        }
        const float y_sum = (float)ysum;
        logp_after[i] = given ? 0.0f : logf(prob_before) - logf(y_sum);
    }
    /* children[action.get_combined_hash()] = CNode(prior, ...): one node per distinct key; key = per-dimension
     * std::to_string(value) == "%f" of the value */
    n->first_child = t->n_nodes[env];
    t->n_nodes[env] += K;
    int distinct = 0;
    for (int i = 0; i < K; ++i) {
        rep[i] = i;
        for (int i2 = 0; i2 < i; ++i2) {
            int same = 1;
            for (int j = 0; j < D && same; ++j) {
                char a[64], b[64];
                snprintf(a, sizeof a, "%f", (double)act[i * D + j]);
                snprintf(b, sizeof b, "%f", (double)act[i2 * D + j]);
                if (strcmp(a, b) != 0) same = 0;
            }
            if (same) { rep[i] = rep[i2]; break; }
        }
        SNode *c = &pool[n->first_child + rep[i]];
        memset(c, 0, sizeof(*c)); /* a later duplicate re-assigns a fresh CNode to the same key */
        c->prior = logp_after[i];
        c->best_slot = -1; c->latent_index = -1; c->batch_index = -1;
        if (rep[i] == i) distinct++;
    }
    n->n_children = distinct;
    n->expanded = 1;
}

/* CRoots::prepare / prepare_no_noise (cnode.cpp:671-702); add_exploration_noise (:454-479, log-prob form) */
void stree_prepare(STree *t, float noise_w, const float *noises /* [B][K] or NULL */, const float *value_prefixs,
                   const float *policies /* [B][2D] */, const int *to_play, const float *given /* [B][K][D] or NULL */)
{
    for (int i = 0; i < t->B; ++i) {
        SNode *pool = &t->nodes[(size_t)i * t->cap];
        node_expand(t, i, 0, to_play[i], 0, i, value_prefixs[i], policies + (size_t)i * t->pstride,
                    given ? given + (size_t)i * t->K * t->D : NULL);
        if (noises) {
            const int *rep = &t->rep[(((size_t)i * t->cap_exp) + 0) * t->K];
            for (int k = 0; k < t->K; ++k) {
                SNode *child = &pool[pool[0].first_child + rep[k]];
                const float prior = child->prior, noise = noises[(size_t)i * t->K + k];
                if (t->A_disc) child->prior = prior * (1 - noise_w) + noise * noise_w;  /* prior is a probability (:474-477) */
                else child->prior = (float)log((double)(expf(prior) * (1 - noise_w) + noise * noise_w) + 1e-6);
            }
        }
        pool[0].visit_count += 1;
    }
}

static const int *node_rep(const STree *t, int env, int ni)
{
    return &t->rep[(((size_t)env * t->cap_exp) + t->exp_of_node[(size_t)env * t->cap + ni]) * t->K];
}

static float compute_mean_q(const STree *t, int env, int ni, int is_root, float parent_q, float discount) /* :480-520 */
{
    const SNode *pool = &t->nodes[(size_t)env * t->cap];
    const SNode *n = &pool[ni];
    const int *rep = node_rep(t, env, ni);
    float total_unsigned_q = 0.0f;
    int total_visits = 0;
    const float parent_value_prefix = n->value_prefix;
    for (int k = 0; k < t->K; ++k) {
        const SNode *child = &pool[n->first_child + rep[k]];
        if (child->visit_count > 0) {
            float true_reward = child->value_prefix - parent_value_prefix;
            if (n->is_reset == 1) true_reward = child->value_prefix;
            const float qsa = true_reward + discount * node_value(child);
            total_unsigned_q += qsa;
            total_visits += 1;
        }
    }
    if (is_root && total_visits > 0) return total_unsigned_q / total_visits;
    return (parent_q + total_unsigned_q) / (total_visits + 1);
}

static float ucb_score(const SNode *parent, const SNode *child, const SMinMax *mm, float parent_mean_q, int is_reset,
                       float total_children_visit_counts, float parent_value_prefix, float pb_c_base, float pb_c_init,
                       float discount, int players) /* cucb_score :1026-1108, the shipped "uniform" branch */
{
    float pb_c, prior_score, value_score = 0.0f;
    pb_c = logf((total_children_visit_counts + pb_c_base + 1) / pb_c_base) + pb_c_init;
    pb_c *= (sqrtf(total_children_visit_counts) / (child->visit_count + 1));
    prior_score = pb_c * 1 / (float)parent->n_children;
    if (child->visit_count == 0) {
        value_score = parent_mean_q;
    } else {
        float true_reward = child->value_prefix - parent_value_prefix;
        if (is_reset == 1) true_reward = child->value_prefix;
        if (players == 1) value_score = true_reward + discount * node_value(child);
        else if (players == 2) value_score = true_reward + discount * (-node_value(child));
    }
    value_score = mm_normalize(mm, value_score);
    if (value_score < 0) value_score = 0;
    if (value_score > 1) value_score = 1;
    return prior_score + value_score;
}

static int select_child(const STree *t, int env, int ni, const SMinMax *mm, int pb_c_base, float pb_c_init,
                        float discount, float mean_q, int players) /* cselect_child :968-1024 -> legal position */
{
    const SNode *pool = &t->nodes[(size_t)env * t->cap];
    const SNode *n = &pool[ni];
    const int *rep = node_rep(t, env, ni);
    float max_score = S_FLOAT_MIN;
    const float epsilon = 0.000001f;
    int lst[t->K], n_max = 0;
    for (int k = 0; k < t->K; ++k) {
        const SNode *child = &pool[n->first_child + rep[k]];
        const float s = ucb_score(n, child, mm, mean_q, n->is_reset, (float)(n->visit_count - 1), n->value_prefix,
                                  (float)pb_c_base, pb_c_init, discount, players);
        if (max_score < s) { max_score = s; n_max = 0; lst[n_max++] = k; }
        else if (s >= max_score - epsilon) lst[n_max++] = k;
    }
    if (n_max == 0) return -1;
    return lst[t->tiebreak ? (rand() % n_max) : 0];
}

void stree_traverse(STree *t, int pb_c_base, float pb_c_init, float discount, int *virtual_to_play, int *out_ix,
                    int *out_iy, float *out_last_action /* [B][D] */, int *out_search_len) /* :1110-1187 */
{
    int largest = virtual_to_play[0];
    for (int i = 1; i < t->B; ++i) if (virtual_to_play[i] > largest) largest = virtual_to_play[i];
    const int players = (largest == -1) ? 1 : 2;
    for (int i = 0; i < t->B; ++i) {
        SNode *pool = &t->nodes[(size_t)i * t->cap];
        int *path = &t->path[(size_t)i * t->cap];
        float parent_q = 0.0f;
        int ni = 0, is_root = 1, search_len = 0, plen = 0;
        path[plen++] = ni;
        while (pool[ni].expanded) {
            SNode *node = &pool[ni];
            const float mean_q = compute_mean_q(t, i, ni, is_root, parent_q, discount);
            is_root = 0;
            parent_q = mean_q;
            const int k = select_child(t, i, ni, &t->mm[i], pb_c_base, pb_c_init, discount, mean_q, players);
            if (players > 1) virtual_to_play[i] = (virtual_to_play[i] == 1) ? 2 : 1;
            node->best_slot = k;
            const int e = t->exp_of_node[(size_t)i * t->cap + ni];
            const float *act = &t->actions[((((size_t)i * t->cap_exp) + e) * t->K + k) * t->D];
            for (int j = 0; j < t->D; ++j) out_last_action[(size_t)i * t->D + j] = act[j];
            ni = node->first_child + node_rep(t, i, ni)[k];
            path[plen++] = ni;
            search_len += 1;
        }
        const SNode *parent = &pool[path[plen - 2]];
        out_ix[i] = parent->latent_index;
        out_iy[i] = parent->batch_index;
        out_search_len[i] = search_len;
        t->path_len[i] = plen;
    }
}

static void backpropagate(STree *t, int env, int to_play, float value, float discount) /* :860-945 */
{
    SNode *pool = &t->nodes[(size_t)env * t->cap];
    const int *path = &t->path[(size_t)env * t->cap];
    SMinMax *mm = &t->mm[env];
    float bootstrap_value = value;
    for (int i = t->path_len[env] - 1; i >= 0; --i) {
        SNode *node = &pool[path[i]];
        if (to_play == -1 || node->to_play == to_play) node->value_sum += bootstrap_value;
        else node->value_sum += -bootstrap_value;
        node->visit_count += 1;
        float parent_value_prefix = 0.0f;
        int is_reset = 0;
        if (i >= 1) { parent_value_prefix = pool[path[i - 1]].value_prefix; is_reset = pool[path[i - 1]].is_reset; }
        float true_reward = node->value_prefix - parent_value_prefix;
        mm_update(mm, true_reward + discount * node_value(node));
        if (is_reset == 1) true_reward = node->value_prefix;
        if (to_play == -1) bootstrap_value = true_reward + discount * bootstrap_value;
        else if (node->to_play == to_play) bootstrap_value = -true_reward + discount * bootstrap_value;
        else bootstrap_value = true_reward + discount * bootstrap_value;
    }
}

void stree_backpropagate(STree *t, int latent_index, float discount, const float *value_prefixs, const float *values,
                         const float *policies, const int *is_reset, const int *to_play, const float *given) /* :947-966 */
{
    for (int i = 0; i < t->B; ++i) {
        const int leaf = t->path[(size_t)i * t->cap + t->path_len[i] - 1];
        node_expand(t, i, leaf, to_play[i], latent_index, i, value_prefixs[i], policies + (size_t)i * t->pstride,
                    given ? given + (size_t)i * t->K * t->D : NULL);
        t->nodes[(size_t)i * t->cap + leaf].is_reset = is_reset[i];
        backpropagate(t, i, to_play[i], values[i], discount);
    }
}

void stree_get_distributions(const STree *t, int *out /* [B][K] */)
{
    for (int i = 0; i < t->B; ++i) {
        const SNode *pool = &t->nodes[(size_t)i * t->cap];
        const int *rep = node_rep(t, i, 0);
        for (int k = 0; k < t->K; ++k) out[(size_t)i * t->K + k] = pool[pool[0].first_child + rep[k]].visit_count;
    }
}
void stree_get_values(const STree *t, float *out) { for (int i = 0; i < t->B; ++i) out[i] = node_value(&t->nodes[(size_t)i * t->cap]); }
void stree_get_minmax(const STree *t, float *out) { for (int i = 0; i < t->B; ++i) { out[2 * i] = t->mm[i].minimum; out[2 * i + 1] = t->mm[i].maximum; } }

/* sampled actions of the node expanded as record `e` of root `env` (e = 0: the root; e = s+1: simulation s) */
void stree_get_actions(const STree *t, int e, float *out /* [B][K][D] */)
{
    for (int i = 0; i < t->B; ++i)
        memcpy(out + (size_t)i * t->K * t->D, &t->actions[(((size_t)i * t->cap_exp) + e) * t->K * t->D],
               sizeof(float) * t->K * t->D);
}
