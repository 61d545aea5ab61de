/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement (plain C11, scalar, single thread) of the reference's
 * batched MCTS trees.  Nothing on the product path may include, link or call this file; it is
 * used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker.
 *
 * Restates (reference = /root/reference, LightZero v0.2.0):
 *   variant 0 "EZ"  lzero/mcts/ctree/ctree_efficientzero/lib/cnode.cpp   (value-prefix tree)
 *   variant 1 "MZ"  lzero/mcts/ctree/ctree_muzero/lib/cnode.cpp          (reward tree)
 *   min-max stats   lzero/mcts/ctree/common_lib/cminimax.cpp:6-45
 *
 * Tie-breaking: the reference draws rand() % n over its tie list after srand(tv_usec)
 * (cnode.cpp:691, utils.cpp:23-25).  `tiebreak` = 0 restates the deterministic build of the
 * reference (rand() -> 0, identical to mz_tree's own deterministic=True, ctree_muzero
 * cnode.cpp:592): front of the tie list.  `tiebreak` = 1 calls rand() like the reference.
 *
 * Parity pin: tests/test_oracle_vs_reference.py drives this file and the compiled reference
 * (oracle/_ref/det, built by oracle/build_ref.py from the reference's own sources) with the same
 * inputs and requires identical per-simulation (ix, iy, last_action, search_len), visit counts,
 * root values and min/max stats; tests/golden/ holds vectors generated from the compiled
 * reference for the GPU box, where /root/reference does not exist.
 *
 * All arithmetic is IEEE binary32 with the same operation order as the C++ (`float` variables,
 * float overloads of exp/log/sqrt = expf/logf/sqrtf of the host libm); compile with
 * -ffp-contract=off (x86-64 baseline g++ emits no FMA for the reference either).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define O_FLOAT_MAX 1000000.0f /* cminimax.h:9 */
#define O_FLOAT_MIN (-O_FLOAT_MAX)

typedef struct {
    int visit_count, to_play, latent_index, batch_index, best_action, is_reset; /* cnode.h:22 */
    float value_prefix; /* EZ: value_prefix; MZ: reward (ctree_muzero/lib/cnode.h) */
    float prior, value_sum;
    int expanded;     /* children.size() > 0, cnode.cpp:214-221 */
    int first_child;  /* index into env node pool of child for action 0 (A consecutive slots) */
    int n_legal;      /* legal_actions.size() */
    int legal_off;    /* offset into env legal list; -1 => 0..A-1 */
} ONode;

typedef struct {
    float maximum, minimum, value_delta_max; /* cminimax.cpp:6-10 */
} OMinMax;

typedef struct {
    int variant, B, A, cap, tiebreak;
    ONode *nodes;   /* [B][cap] */
    int *n_nodes;   /* [B] */
    int *legal;     /* [B][A] root legal list (order preserved) */
    int *n_legal;   /* [B] */
    OMinMax *mm;    /* [B] */
    int *path;      /* [B][cap] node indices of search path (CSearchResults.search_paths) */
    int *path_len;  /* [B] */
} OTree;

static void node_init(ONode *n, float prior) /* cnode.cpp:63-84 */
{
    n->prior = prior;
    n->is_reset = 0;
    n->visit_count = 0;
    n->value_sum = 0;
    n->best_action = -1;
    n->to_play = 0;
    n->value_prefix = 0.0f;
    n->latent_index = -1;
    n->batch_index = -1;
    n->expanded = 0;
    n->first_child = -1;
    n->n_legal = 0;
    n->legal_off = -1;
}

OTree *otree_create(int variant, int B, int A, int max_sims, const int *legal_flat, const int *legal_cnt)
{
    /* CRoots::CRoots cnode.cpp:305-321 : one root CNode(0, legal_actions_list[i]) per env */
    OTree *t = (OTree *)calloc(1, sizeof(OTree));
    t->variant = variant; t->B = B; t->A = A; t->tiebreak = 0;
    t->cap = 1 + (max_sims + 1) * A; /* root + A children per expansion (root + one per simulation) */
    t->nodes = (ONode *)malloc(sizeof(ONode) * (size_t)B * t->cap);
    t->n_nodes = (int *)calloc(B, sizeof(int));
    t->legal = (int *)calloc((size_t)B * A, sizeof(int));
    t->n_legal = (int *)calloc(B, sizeof(int));
    t->mm = (OMinMax *)malloc(sizeof(OMinMax) * B);
    t->path = (int *)calloc((size_t)B * t->cap, sizeof(int));
    t->path_len = (int *)calloc(B, sizeof(int));
    int off = 0;
    for (int i = 0; i < B; ++i) {
        t->n_legal[i] = legal_cnt ? legal_cnt[i] : 0;
        for (int j = 0; j < t->n_legal[i]; ++j) t->legal[(size_t)i * A + j] = legal_flat[off + j];
        off += t->n_legal[i];
        ONode *root = &t->nodes[(size_t)i * t->cap];
        node_init(root, 0.0f);
        root->n_legal = t->n_legal[i];
        root->legal_off = 0;
        t->n_nodes[i] = 1;
        t->mm[i].maximum = O_FLOAT_MIN; t->mm[i].minimum = O_FLOAT_MAX; t->mm[i].value_delta_max = 0.0f;
    }
    return t;
}

void otree_destroy(OTree *t)
{
    if (!t) return;
    free(t->nodes); free(t->n_nodes); free(t->legal); free(t->n_legal); free(t->mm); free(t->path); free(t->path_len);
    free(t);
}

void otree_set_tiebreak(OTree *t, int mode) { t->tiebreak = mode; }

void otree_set_delta(OTree *t, float d) /* CMinMaxStatsList::set_delta cminimax.cpp:61-65 */
{
    for (int i = 0; i < t->B; ++i) t->mm[i].value_delta_max = d;
}

static void mm_update(OMinMax *m, float v) /* cminimax.cpp:19-26 */
{
    if (v > m->maximum) m->maximum = v;
    if (v < m->minimum) m->minimum = v;
}

static float mm_normalize(const OMinMax *m, float value) /* cminimax.cpp:33-45 */
{
    float norm_value = value;
    float delta = m->maximum - m->minimum;
    if (delta > 0) {
        if (delta < m->value_delta_max) norm_value = (norm_value - m->minimum) / m->value_delta_max;
        else norm_value = (norm_value - m->minimum) / delta;
    }
    return norm_value;
}

static int legal_at(const OTree *t, int env, const ONode *n, int j)
{
    return n->legal_off < 0 ? j : t->legal[(size_t)env * t->A + j];
}

static float node_value(const ONode *n) /* cnode.cpp:223-239 */
{
    if (n->visit_count == 0) return 0.0f;
    return n->value_sum / n->visit_count;
}

/* CNode::expand  cnode.cpp:88-151 (EZ) / ctree_muzero cnode.cpp:83-147 (MZ) */
static void node_expand(OTree *t, int env, int ni, int to_play, int latent_index, int batch_index,
                        float value_prefix, const float *logits)
{
    ONode *pool = &t->nodes[(size_t)env * t->cap];
    ONode *n = &pool[ni];
    const int A = t->A;
    n->to_play = to_play;
    n->latent_index = latent_index;
    n->batch_index = batch_index;
    n->value_prefix = value_prefix;
    if (n->n_legal == 0) { n->n_legal = A; n->legal_off = -1; } /* :106-112 all actions */
    float policy[A];
    float policy_sum = 0.0f;
    float policy_max = O_FLOAT_MIN;
    for (int j = 0; j < n->n_legal; ++j) {
        int a = legal_at(t, env, n, j);
        if (policy_max < logits[a]) policy_max = logits[a];
    }
    for (int j = 0; j < n->n_legal; ++j) {
        int a = legal_at(t, env, n, j);
        float temp_policy = expf(logits[a] - policy_max); /* exp(float) -> float overload */
        policy_sum += temp_policy;
        policy[a] = temp_policy;
    }
    n->first_child = t->n_nodes[env];
    t->n_nodes[env] += A;
    for (int a = 0; a < A; ++a) node_init(&pool[n->first_child + a], 0.0f);
    for (int j = 0; j < n->n_legal; ++j) {
        int a = legal_at(t, env, n, j);
        pool[n->first_child + a].prior = policy[a] / policy_sum;
    }
    n->expanded = 1;
}

/* CRoots::prepare / prepare_no_noise  cnode.cpp:325-360 ; add_exploration_noise :153-171 */
void otree_prepare(OTree *t, float noise_w, const float *noises_flat, const float *value_prefixs,
                   const float *logits, const int *to_play)
{
    int off = 0;
    for (int i = 0; i < t->B; ++i) {
        ONode *pool = &t->nodes[(size_t)i * t->cap];
        node_expand(t, i, 0, to_play[i], 0, i, value_prefixs[i], logits + (size_t)i * t->A);
        ONode *root = &pool[0];
        if (noises_flat) {
            for (int j = 0; j < root->n_legal; ++j) {
                float noise = noises_flat[off + j];
                ONode *child = &pool[root->first_child + legal_at(t, i, root, j)];
                float prior = child->prior;
                child->prior = prior * (1 - noise_w) + noise * noise_w;
            }
            off += root->n_legal;
        }
        root->visit_count += 1;
    }
}

/* CNode::compute_mean_q  cnode.cpp:173-212 (EZ) / ctree_muzero cnode.cpp:169-203 (MZ) */
static float compute_mean_q(const OTree *t, int env, const ONode *n, int is_root, float parent_q, float discount)
{
    const ONode *pool = &t->nodes[(size_t)env * t->cap];
    float total_unsigned_q = 0.0f;
    int total_visits = 0;
    float parent_value_prefix = n->value_prefix;
    for (int j = 0; j < n->n_legal; ++j) {
        const ONode *child = &pool[n->first_child + legal_at(t, env, n, j)];
        if (child->visit_count > 0) {
            float true_reward;
            if (t->variant == 0) {
                true_reward = child->value_prefix - parent_value_prefix;
                if (n->is_reset == 1) true_reward = child->value_prefix;
            } else {
                true_reward = child->value_prefix; /* MZ: child->reward */
            }
            float qsa = true_reward + discount * node_value(child);
            total_unsigned_q += qsa;
            total_visits += 1;
        }
    }
    float mean_q;
    if (is_root && total_visits > 0) mean_q = total_unsigned_q / total_visits;
    else mean_q = (parent_q + total_unsigned_q) / (total_visits + 1);
    return mean_q;
}

/* cucb_score  cnode.cpp:756-814 (EZ) / ctree_muzero cnode.cpp:654-698 (MZ) */
static float ucb_score(const OTree *t, const ONode *child, const OMinMax *mm, float parent_mean_q, int is_reset,
                       float total_children_visit_counts, float parent_value_prefix, float pb_c_base,
                       float pb_c_init, float discount, int players)
{
    float pb_c = 0.0f, prior_score = 0.0f, value_score = 0.0f;
    pb_c = logf((total_children_visit_counts + pb_c_base + 1) / pb_c_base) + pb_c_init;
    pb_c *= (sqrtf(total_children_visit_counts) / (child->visit_count + 1));
    prior_score = pb_c * child->prior;
    if (child->visit_count == 0) {
        value_score = parent_mean_q;
    } else {
        float true_reward;
        if (t->variant == 0) {
            true_reward = child->value_prefix - parent_value_prefix;
            if (is_reset == 1) true_reward = child->value_prefix;
        } else {
            true_reward = child->value_prefix;
        }
        if (players == 1) value_score = true_reward + discount * node_value(child);
        else if (players == 2) value_score = true_reward + discount * (-node_value(child));
    }
    value_score = mm_normalize(mm, value_score);
    if (value_score < 0) value_score = 0;
    else if (value_score > 1) value_score = 1;
    return prior_score + value_score;
}

/* cselect_child  cnode.cpp:651-695 (EZ) / ctree_muzero cnode.cpp:551-595 (MZ) */
static int select_child(const OTree *t, int env, const ONode *n, const OMinMax *mm, int pb_c_base, float pb_c_init,
                        float discount, float mean_q, int players)
{
    const ONode *pool = &t->nodes[(size_t)env * t->cap];
    float max_score = O_FLOAT_MIN;
    const float epsilon = 0.000001f;
    int max_index_lst[t->A];
    int n_max = 0;
    for (int j = 0; j < n->n_legal; ++j) {
        int a = legal_at(t, env, n, j);
        const ONode *child = &pool[n->first_child + a];
        float temp_score = ucb_score(t, child, mm, mean_q, n->is_reset, (float)(n->visit_count - 1), n->value_prefix,
                                     (float)pb_c_base, pb_c_init, discount, players);
        if (max_score < temp_score) {
            max_score = temp_score;
            n_max = 0;
            max_index_lst[n_max++] = a;
        } else if (temp_score >= max_score - epsilon) {
            max_index_lst[n_max++] = a;
        }
    }
    int action = 0;
    if (n_max > 0) {
        int rand_index = t->tiebreak ? (rand() % n_max) : 0;
        action = max_index_lst[rand_index];
    }
    return action;
}

/* cbatch_traverse  cnode.cpp:886-963 (EZ) / ctree_muzero cnode.cpp:754-825 (MZ) */
void otree_traverse(OTree *t, int pb_c_base, float pb_c_init, float discount, int *virtual_to_play, int *out_ix,
                    int *out_iy, int *out_last_action, int *out_search_len)
{
    int last_action = -1;
    int players;
    int largest = virtual_to_play[0];
    for (int i = 1; i < t->B; ++i) if (virtual_to_play[i] > largest) largest = virtual_to_play[i];
    players = (largest == -1) ? 1 : 2;
    for (int i = 0; i < t->B; ++i) {
        ONode *pool = &t->nodes[(size_t)i * t->cap];
        int *path = &t->path[(size_t)i * t->cap];
        float parent_q = 0.0f;
        int ni = 0, is_root = 1, search_len = 0, plen = 0;
        path[plen++] = ni;
        while (pool[ni].expanded) {
            ONode *node = &pool[ni];
            float mean_q = compute_mean_q(t, i, node, is_root, parent_q, discount);
            is_root = 0;
            parent_q = mean_q;
            int action = select_child(t, i, node, &t->mm[i], pb_c_base, pb_c_init, discount, mean_q, players);
            if (players > 1) virtual_to_play[i] = (virtual_to_play[i] == 1) ? 2 : 1;
            node->best_action = action;
            ni = node->first_child + action;
            last_action = action;
            path[plen++] = ni;
            search_len += 1;
        }
        const ONode *parent = &pool[path[plen - 2]];
        out_ix[i] = parent->latent_index;
        out_iy[i] = parent->batch_index;
        out_last_action[i] = last_action;
        out_search_len[i] = search_len;
        t->path_len[i] = plen;
    }
}

/* Diagnostic (no counterpart in the reference): the selection the NEXT otree_traverse would make for root `env`, level by level,
 * WITHOUT touching the tree -- per level the node's latent index, the cucb_score of every legal action (action-indexed row of A
 * floats, illegal = O_FLOAT_MIN) and the action the deterministic rule picks.  Used by tests/test_e2e_cfg1_gpu.py to attribute a root
 * whose visit counts differ between two pipelines to the first selection that differed and to say how close its best two scores were.
 * Returns the number of levels written (<= max_levels). */
int otree_probe(const OTree *t, int env, int pb_c_base, float pb_c_init, float discount, int players, int max_levels,
                int *out_node_latent, float *out_scores, int *out_action)
{
    const ONode *pool = &t->nodes[(size_t)env * t->cap];
    float parent_q = 0.0f;
    int ni = 0, is_root = 1, lv = 0;
    while (pool[ni].expanded && lv < max_levels) {
        const ONode *node = &pool[ni];
        float mean_q = compute_mean_q(t, env, node, is_root, parent_q, discount);
        is_root = 0;
        parent_q = mean_q;
        float *row = out_scores + (size_t)lv * t->A;
        for (int a = 0; a < t->A; ++a) row[a] = O_FLOAT_MIN;
        float max_score = O_FLOAT_MIN;
        int best = 0, have = 0;
        for (int j = 0; j < node->n_legal; ++j) {
            int a = legal_at(t, env, node, j);
            const ONode *child = &pool[node->first_child + a];
            float sc = ucb_score(t, child, &t->mm[env], mean_q, node->is_reset, (float)(node->visit_count - 1), node->value_prefix,
                                 (float)pb_c_base, pb_c_init, discount, players);
            row[a] = sc;
            if (!have || max_score < sc) { max_score = sc; best = a; have = 1; }  /* front of the tie list: first maximum */
        }
        out_node_latent[lv] = node->latent_index;
        out_action[lv] = best;
        ni = node->first_child + best;
        ++lv;
    }
    return lv;
}

/* cbackpropagate  cnode.cpp:482-575 (EZ) / ctree_muzero cnode.cpp:419-478 (MZ) */
static void backpropagate(OTree *t, int env, int to_play, float value, float discount)
{
    ONode *pool = &t->nodes[(size_t)env * t->cap];
    const int *path = &t->path[(size_t)env * t->cap];
    OMinMax *mm = &t->mm[env];
    float bootstrap_value = value;
    for (int i = t->path_len[env] - 1; i >= 0; --i) {
        ONode *node = &pool[path[i]];
        if (to_play == -1 || node->to_play == to_play) node->value_sum += bootstrap_value;
        else node->value_sum += -bootstrap_value;
        node->visit_count += 1;
        float true_reward;
        int is_reset = 0;
        if (t->variant == 0) {
            float parent_value_prefix = 0.0f;
            if (i >= 1) {
                const ONode *parent = &pool[path[i - 1]];
                parent_value_prefix = parent->value_prefix;
                is_reset = parent->is_reset;
            }
            true_reward = node->value_prefix - parent_value_prefix;
            mm_update(mm, true_reward + discount * node_value(node)); /* :516 / :558, before the reset override */
            if (is_reset == 1) true_reward = node->value_prefix;
        } else {
            true_reward = node->value_prefix; /* node->reward */
            if (to_play == -1) mm_update(mm, true_reward + discount * node_value(node));     /* mz :443 */
            else mm_update(mm, true_reward + discount * -node_value(node));                  /* mz :470 */
        }
        if (to_play == -1) bootstrap_value = true_reward + discount * bootstrap_value;
        else if (node->to_play == to_play) bootstrap_value = -true_reward + discount * bootstrap_value;
        else bootstrap_value = true_reward + discount * bootstrap_value;
    }
}

/* cbatch_backpropagate  cnode.cpp:577-601 (EZ) / ctree_muzero (MZ, no is_reset_list) */
void otree_backpropagate(OTree *t, int latent_index, float discount, const float *value_prefixs, const float *values,
                         const float *logits, const int *is_reset, const int *to_play)
{
    for (int i = 0; i < t->B; ++i) {
        int leaf = t->path[(size_t)i * t->cap + t->path_len[i] - 1];
        node_expand(t, i, leaf, to_play[i], latent_index, i, value_prefixs[i], logits + (size_t)i * t->A);
        if (t->variant == 0) t->nodes[(size_t)i * t->cap + leaf].is_reset = is_reset[i];
        backpropagate(t, i, to_play[i], values[i], discount);
    }
}

/* ---- ReZero: search_with_reuse (https://arxiv.org/abs/2404.16364) ------------------------------------------------ */
/* carm_score  cnode.cpp:816-884 (EZ) / ctree_muzero cnode.cpp:701-752 (MZ) */
static float arm_score(const OTree *t, const ONode *child, const OMinMax *mm, float parent_mean_q, int is_reset,
                       float reuse_value, float total_children_visit_counts, float parent_value_prefix, float pb_c_base,
                       float pb_c_init, float discount, int players)
{
    float pb_c = 0.0f, prior_score = 0.0f, value_score = 0.0f;
    pb_c = logf((total_children_visit_counts + pb_c_base + 1) / pb_c_base) + pb_c_init;
    pb_c *= (sqrtf(total_children_visit_counts) / (child->visit_count + 1));
    prior_score = pb_c * child->prior;
    if (child->visit_count == 0) {
        value_score = parent_mean_q;
    } else {
        float true_reward;
        if (t->variant == 0) {
            true_reward = child->value_prefix - parent_value_prefix;
            if (is_reset == 1) true_reward = child->value_prefix;
        } else {
            true_reward = child->value_prefix;
        }
        if (players == 1) value_score = true_reward + discount * reuse_value;
        else if (players == 2) value_score = true_reward + discount * (-reuse_value);
    }
    value_score = mm_normalize(mm, value_score);
    if (value_score < 0) value_score = 0;
    else if (value_score > 1) value_score = 1;
    return (child->visit_count == 0) ? prior_score + value_score : value_score;
}

/* cselect_root_child  cnode.cpp:697-754 (EZ) / ctree_muzero cnode.cpp:597-652 (MZ) */
static int select_root_child(const OTree *t, int env, const ONode *n, const OMinMax *mm, int pb_c_base, float pb_c_init,
                             float discount, float mean_q, int players, int true_action, float reuse_value)
{
    const ONode *pool = &t->nodes[(size_t)env * t->cap];
    float max_score = O_FLOAT_MIN;
    const float epsilon = 0.000001f;
    int max_index_lst[t->A];
    int n_max = 0;
    for (int j = 0; j < n->n_legal; ++j) {
        int a = legal_at(t, env, n, j);
        const ONode *child = &pool[n->first_child + a];
        float temp_score;
        if (a == true_action)
            temp_score = arm_score(t, child, mm, mean_q, n->is_reset, reuse_value, (float)(n->visit_count - 1), n->value_prefix,
                                   (float)pb_c_base, pb_c_init, discount, players);
        else
            temp_score = ucb_score(t, child, mm, mean_q, n->is_reset, (float)(n->visit_count - 1), n->value_prefix,
                                   (float)pb_c_base, pb_c_init, discount, players);
        if (max_score < temp_score) {
            max_score = temp_score;
            n_max = 0;
            max_index_lst[n_max++] = a;
        } else if (temp_score >= max_score - epsilon) {
            max_index_lst[n_max++] = a;
        }
    }
    int action = 0;
    if (n_max > 0) {
        int rand_index = t->tiebreak ? (rand() % n_max) : 0;
        action = max_index_lst[rand_index];
    }
    return action;
}

/* cbatch_traverse_with_reuse  cnode.cpp:965-1072 (EZ) / ctree_muzero cnode.cpp:828-930 (MZ):
 * the search stops below the root when the root picks the trajectory's true action; out_ix = -1 when the node reached
 * is already expanded (no inference needed for this root in this simulation). */
void otree_traverse_with_reuse(OTree *t, int pb_c_base, float pb_c_init, float discount, int *virtual_to_play,
                               const int *true_action, const float *reuse_value, int *out_ix, int *out_iy,
                               int *out_last_action, int *out_search_len)
{
    int last_action = -1;
    int players;
    int largest = virtual_to_play[0];
    for (int i = 1; i < t->B; ++i) if (virtual_to_play[i] > largest) largest = virtual_to_play[i];
    players = (largest == -1) ? 1 : 2;
    for (int i = 0; i < t->B; ++i) {
        ONode *pool = &t->nodes[(size_t)i * t->cap];
        int *path = &t->path[(size_t)i * t->cap];
        float parent_q = 0.0f;
        int ni = 0, is_root = 1, search_len = 0, plen = 0;
        path[plen++] = ni;
        while (pool[ni].expanded) {
            ONode *node = &pool[ni];
            float mean_q = compute_mean_q(t, i, node, is_root, parent_q, discount);
            parent_q = mean_q;
            int action;
            if (is_root) action = select_root_child(t, i, node, &t->mm[i], pb_c_base, pb_c_init, discount, mean_q, players, true_action[i], reuse_value[i]);
            else action = select_child(t, i, node, &t->mm[i], pb_c_base, pb_c_init, discount, mean_q, players);
            if (players > 1) virtual_to_play[i] = (virtual_to_play[i] == 1) ? 2 : 1;
            node->best_action = action;
            ni = node->first_child + action;
            last_action = action;
            path[plen++] = ni;
            search_len += 1;
            if (is_root && action == true_action[i]) break;
            is_root = 0;
        }
        if (pool[ni].expanded) {
            out_ix[i] = -1;
            out_iy[i] = i;
        } else {
            const ONode *parent = &pool[path[plen - 2]];
            out_ix[i] = parent->latent_index;
            out_iy[i] = parent->batch_index;
        }
        out_last_action[i] = last_action;
        out_search_len[i] = search_len;
        t->path_len[i] = plen;
    }
}

/* cbatch_backpropagate_with_reuse  cnode.cpp:603-649 (EZ) / ctree_muzero cnode.cpp:502-549 (MZ).
 * value_prefixs / values / logits hold one row per root that needed inference, in root order (the reference's compacted
 * batch); no_inference_lst / reuse_lst are ascending root indices terminated by -1. */
void otree_backpropagate_with_reuse(OTree *t, int latent_index, float discount, const float *value_prefixs,
                                    const float *values, const float *logits, const int *is_reset, const int *to_play,
                                    const int *no_inference_lst, const int *reuse_lst, const float *reuse_value)
{
    int count_a = 0, count_b = 0, count_c = 0;
    float value_propagate = 0;
    for (int i = 0; i < t->B; ++i) {
        int leaf = t->path[(size_t)i * t->cap + t->path_len[i] - 1];
        if (i == no_inference_lst[count_a]) {
            count_a = count_a + 1;
            value_propagate = reuse_value[i];
        } else {
            node_expand(t, i, leaf, to_play[i], latent_index, count_b, value_prefixs[count_b], logits + (size_t)count_b * t->A);
            if (i == reuse_lst[count_c]) {
                value_propagate = reuse_value[i];
                count_c = count_c + 1;
            } else {
                value_propagate = values[count_b];
            }
            count_b = count_b + 1;
        }
        if (t->variant == 0) t->nodes[(size_t)i * t->cap + leaf].is_reset = is_reset[i];
        backpropagate(t, i, to_play[i], value_propagate, discount);
    }
}

/* CRoots::get_distributions cnode.cpp:389-405, CNode::get_children_distribution :263-281 */
void otree_get_distributions(const OTree *t, int *out /* [B][A], -1 padded */, int *out_cnt)
{
    for (int i = 0; i < t->B; ++i) {
        const ONode *pool = &t->nodes[(size_t)i * t->cap];
        const ONode *root = &pool[0];
        for (int j = 0; j < t->A; ++j) out[(size_t)i * t->A + j] = -1;
        out_cnt[i] = 0;
        if (!root->expanded) continue;
        for (int j = 0; j < root->n_legal; ++j)
            out[(size_t)i * t->A + j] = pool[root->first_child + legal_at(t, i, root, j)].visit_count;
        out_cnt[i] = root->n_legal;
    }
}

void otree_get_values(const OTree *t, float *out) /* CRoots::get_values cnode.cpp:407-419 */
{
    for (int i = 0; i < t->B; ++i) out[i] = node_value(&t->nodes[(size_t)i * t->cap]);
}

void otree_get_minmax(const OTree *t, float *out /* [B][2] = (min, max) */)
{
    for (int i = 0; i < t->B; ++i) { out[2 * i] = t->mm[i].minimum; out[2 * i + 1] = t->mm[i].maximum; }
}

/* CNode::get_trajectory cnode.cpp:241-261 : follow best_action from the root; out [B][cap] -1 terminated */
void otree_get_trajectories(const OTree *t, int *out, int stride)
{
    for (int i = 0; i < t->B; ++i) {
        const ONode *pool = &t->nodes[(size_t)i * t->cap];
        const ONode *node = &pool[0];
        int k = 0;
        int best = node->best_action;
        while (best >= 0 && k < stride - 1) {
            out[(size_t)i * stride + k++] = best;
            node = &pool[node->first_child + best];
            best = node->best_action;
        }
        out[(size_t)i * stride + k] = -1;
    }
}
