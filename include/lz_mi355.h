/*
 * lz_mi355.h -- C ABI of the MI355X-native batched self-play / MCTS inference engine.
 *
 * This is the drop-in boundary for LightZero's hot path.  Each entry point
 * names the reference interface it replaces (paths relative to the LightZero repository root).
 * Plain C: opaque handles, pointers and sizes only -- no torch types.  Every function returns
 * LZ_OK (0) or a negative lz_status; the message of the last error on the calling thread is
 * available from lz_last_error().  Nothing throws, nothing calls exit().  One HIP stream per
 * engine; calls on one engine / roots handle must be serialised by the caller (the reference's
 * Cython module holds the GIL for the whole call and is not re-entrant either).
 *
 * Pointer conventions: parameters named  h_*  are HOST pointers (copied synchronously, like the
 * reference's list -> std::vector deep copies, ez_tree.pyx:82-91); parameters named  d_*  are
 * DEVICE (HBM) pointers used in place on the engine's stream.
 */
#ifndef LZ_MI355_H
#define LZ_MI355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum lz_status {
    LZ_OK = 0,
    LZ_ERR_INVALID = -1,   /* bad argument / call order */
    LZ_ERR_HIP = -2,       /* HIP runtime failure (message has the hipError string) */
    LZ_ERR_NOMEM = -3,
    LZ_ERR_NODEVICE = -4,  /* no gfx950 device visible: the product path fails loudly, no CPU fallback */
    LZ_ERR_STATE = -5      /* weights missing / not finalized / capacity exceeded */
} lz_status;

typedef enum lz_variant {
    LZ_TREE_EFFICIENTZERO = 0, /* lzero/mcts/ctree/ctree_efficientzero (value-prefix tree, is_reset) */
    LZ_TREE_MUZERO = 1,        /* lzero/mcts/ctree/ctree_muzero (reward tree) */
    LZ_TREE_SAMPLED_EFFICIENTZERO = 2, /* lzero/mcts/ctree/ctree_sampled_efficientzero, continuous actions */
    LZ_TREE_GUMBEL_MUZERO = 3          /* lzero/mcts/ctree/ctree_gumbel_muzero */
} lz_variant;

typedef enum lz_tiebreak {
    LZ_TIE_FIRST = 0,  /* front of the reference's tie list == first arg-max (mz_tree deterministic=True,
                          ctree_muzero/lib/cnode.cpp:592; the rand()->0 build of ctree_efficientzero) */
    LZ_TIE_RANDOM = 1  /* uniform over the reference's tie list (cnode.cpp:668-693), counter-based RNG */
} lz_tiebreak;

typedef struct lz_engine lz_engine; /* device context: stream, weights, workspaces */
typedef struct lz_roots lz_roots;   /* a batch of search trees + their min-max stats + last search results */

const char *lz_last_error(void);
int lz_version(void);
/* number of visible gfx950 devices (0 => every compute entry point returns LZ_ERR_NODEVICE) */
int lz_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * Engine
 * ---------------------------------------------------------------------------------------------- */
/* device_index: HIP device ordinal (LOCAL_RANK in a one-process-per-GPU job). */
int lz_engine_create(int device_index, lz_engine **out);
int lz_engine_destroy(lz_engine *e);
int lz_engine_synchronize(lz_engine *e);
/* the engine's hipStream_t as an opaque pointer (so a caller can order its own work / events on it) */
void *lz_engine_stream(lz_engine *e);

/* ------------------------------------------------------------------------------------------------
 * Trees -- replaces the Cython module surface of ez_tree.pyx / mz_tree.pyx
 * ---------------------------------------------------------------------------------------------- */
/* Roots(root_num, legal_actions_list)            ez_tree.pyx:29-32 -> CRoots::CRoots cnode.cpp:305-321
 * + MinMaxStatsList(num)                         ez_tree.pyx:6-10  -> cminimax.cpp:51-59
 * h_legal_flat: concatenated legal-action lists (order preserved), h_legal_count[root_num] their lengths
 * (a length of 0 means "all actions", cnode.cpp:106-112).  max_simulations bounds the node pool:
 * (max_simulations + 1) expanded nodes x action_space_size edges per root, resident in HBM.
 * action_space_size: up to 65535 for LZ_TREE_MUZERO / LZ_TREE_EFFICIENTZERO (beyond 256 -- Chinese chess has 2086 moves,
 * zoo/board_games/chinese_chess/config/chinese_chess_muzero_bot_mode_config.py:33 -- a node's children are walked in 64-lane chunks,
 * csrc/lz_tree_wide.hip, the *_with_reuse entry points included; the fused lz_search stays at 256 with the engine models), up to 1024 for
 * LZ_TREE_GUMBEL_MUZERO (its kernels' 8- / 16-chunk instances). */
int lz_roots_create(lz_engine *e, int variant, int root_num, int action_space_size, int max_simulations,
                    const int32_t *h_legal_flat, const int32_t *h_legal_count, lz_roots **out);
/* Re-arm an existing batch of trees for the next env-step with new legal-action lists (same root_num and
 * action space): what constructing a fresh Roots does in the reference, without re-allocating the HBM pools. */
int lz_roots_reset(lz_roots *r, const int32_t *h_legal_flat, const int32_t *h_legal_count);
/* The same re-arm for a batch whose root inference (lz_initial_inference) is ALREADY enqueued for this env-step: the
 * legal-action lists may arrive after the representation network was launched, so that the host work that builds them
 * (efficientzero.py:579: one np.nonzero per env) overlaps the network instead of preceding it.  Keeps the root latent /
 * predictions; everything else as lz_roots_reset. */
int lz_roots_reset_keep_inference(lz_roots *r, const int32_t *h_legal_flat, const int32_t *h_legal_count);
int lz_roots_destroy(lz_roots *r);
int lz_roots_num(const lz_roots *r);
/* MinMaxStatsList.set_delta                      ez_tree.pyx:12-14 -> cminimax.cpp:61-65.
 * Also resets min/max to (+FLOAT_MAX, -FLOAT_MAX) like a freshly constructed MinMaxStatsList. */
int lz_roots_minmax_reset(lz_roots *r, float value_delta_max);
int lz_roots_set_tiebreak(lz_roots *r, int mode, uint64_t seed);
/* restart the handle's random streams (stochastic tie-breaks, sampled actions, device-side Dirichlet noise, select_action) from `seed`:
 * also rewinds the device-resident epoch that every prepare advances.  For a handle that is re-armed for a new Roots object. */
int lz_roots_reseed(lz_roots *r, uint64_t seed);

/* Roots.prepare / prepare_no_noise               ez_tree.pyx:34-42 -> CRoots::prepare cnode.cpp:325-360
 * h_noises_flat: per root, one noise per LEGAL action in list order (cnode.cpp:163-170); NULL => no noise.
 * h_value_prefix: EZ value prefixes / MZ rewards [root_num]; h_policy_logits [root_num][A]; h_to_play [root_num]. */
int lz_roots_prepare(lz_roots *r, float root_noise_weight, const float *h_noises_flat, const float *h_value_prefix,
                     const float *h_policy_logits, const int32_t *h_to_play);
/* same with device-resident inputs; d_noises is [root_num][A] indexed by position in the legal list */
int lz_roots_prepare_device(lz_roots *r, float root_noise_weight, const float *d_noises, const float *d_value_prefix,
                            const float *d_policy_logits, const int32_t *d_to_play, int players);

/* batch_traverse(roots, pb_c_base, pb_c_init, discount, minmax, results, virtual_to_play)
 *                                                ez_tree.pyx:107-113 -> cbatch_traverse cnode.cpp:886-963
 *                                                mz_tree.pyx:95-98   -> ctree_muzero cnode.cpp:754-825
 * h_virtual_to_play is in/out (flipped once per level in 2-player mode).  Outputs [root_num] each:
 * latent_state_index_in_search_path, latent_state_index_in_batch, last_actions, search_lens
 * (ResultsWrapper.get_search_len, ez_tree.pyx:22-24). */
int lz_batch_traverse(lz_roots *r, int pb_c_base, float pb_c_init, float discount_factor,
                      int32_t *h_virtual_to_play, int32_t *h_out_index_in_search_path, int32_t *h_out_index_in_batch,
                      int32_t *h_out_last_actions, int32_t *h_out_search_lens);
/* batch_backpropagate(current_latent_state_index, discount, value_prefixs, values, policies, minmax, results,
 *                     is_reset_list, to_play_batch)
 *                                                ez_tree.pyx:82-92 -> cbatch_backpropagate cnode.cpp:577-601
 * h_is_reset is ignored (may be NULL) for LZ_TREE_MUZERO. */
int lz_batch_backpropagate(lz_roots *r, int current_latent_state_index, float discount_factor,
                           const float *h_value_prefixs, const float *h_values, const float *h_policy_logits,
                           const int32_t *h_is_reset, const int32_t *h_to_play);

/* Roots.get_distributions / get_values / get_trajectories   ez_tree.pyx:44-54 -> cnode.cpp:371-419
 * h_out_dist [root_num][A]: child visit counts in legal-list order, -1 padded; h_out_count [root_num]. */
int lz_roots_get_distributions(lz_roots *r, int32_t *h_out_dist, int32_t *h_out_count);
int lz_roots_get_values(lz_roots *r, float *h_out_values);
/* h_out [root_num][stride]: best-action chain from each root, -1 terminated */
int lz_roots_get_trajectories(lz_roots *r, int32_t *h_out, int stride);
/* (minimum, maximum) per root [root_num][2] -- observability for tests */
int lz_roots_get_minmax(lz_roots *r, float *h_out);
/* priors of the root's edges after prepare, [root_num][A] by action, 0 where illegal (CNode::prior, cnode.cpp:139-150, 163-170) --
 * observability for tests */
int lz_roots_get_root_priors(lz_roots *r, float *h_out);

/* ------------------------------------------------------------------------------------------------
 * Sampled EfficientZero trees, continuous action space -- replaces ezs_tree.pyx (ctree_sampled_efficientzero).
 * Every expanded node owns K actions drawn from tanh(N(mu, sigma)) (cnode.cpp:238-282).  The draws are made on the
 * device with a counter-based RNG, or -- for bit-exact parity runs -- supplied by the caller (h_given [root_num][K][D]).
 * The UCB prior is the shipped "uniform" branch pb_c / children.size() (cnode.cpp:1054-1079); actions whose every
 * dimension prints identically with "%f" share one child (the reference keys children by a hash of to_string).
 * lz_roots_get_values / lz_roots_minmax_reset / lz_roots_set_tiebreak / lz_roots_destroy apply to these handles too.
 * ---------------------------------------------------------------------------------------------- */
int lz_sroots_create(lz_engine *e, int root_num, int action_dim, int num_of_sampled_actions, int max_simulations,
                     lz_roots **out);
/* Roots.prepare / prepare_no_noise  ezs_tree.pyx -> CRoots::prepare cnode.cpp:671-702.  h_policy [root_num][2D] = (mu | sigma).
 * h_noises [root_num][K] is accepted for signature parity; it only perturbs priors that the shipped score never reads. */
int lz_sroots_prepare(lz_roots *r, float root_noise_weight, const float *h_noises, const float *h_value_prefix,
                      const float *h_policy, const int32_t *h_to_play, const float *h_given);
/* batch_traverse  -> cbatch_traverse cnode.cpp:1110-1187; h_out_last_actions [root_num][D] floats */
int lz_sbatch_traverse(lz_roots *r, int pb_c_base, float pb_c_init, float discount_factor, int32_t *h_virtual_to_play,
                       int32_t *h_out_index_in_search_path, int32_t *h_out_index_in_batch, float *h_out_last_actions,
                       int32_t *h_out_search_lens);
/* batch_backpropagate -> cbatch_backpropagate cnode.cpp:947-966 */
int lz_sbatch_backpropagate(lz_roots *r, int current_latent_state_index, float discount_factor, const float *h_value_prefixs,
                            const float *h_values, const float *h_policy, const int32_t *h_is_reset,
                            const int32_t *h_to_play, const float *h_given);
/* discrete action spaces (continuous_action_space = False, cnode.cpp:288-327): each node samples K of the action_space_size
 * actions without replacement from its policy logits; an action is the float of its index, policies are [root_num][A] logits.
 * action_space_size <= 256 (bipedalwalker_cont_disc_sampled_efficientzero_config.py: 4^4), num_of_sampled_actions <= 64. */
int lz_sroots_create_discrete(lz_engine *e, int root_num, int action_space_size, int num_of_sampled_actions,
                              int max_simulations, lz_roots **out);
/* parity runs of the fused sampled search: inject the post-tanh draws of every expansion, [records][root_num][K][D]
 * (record 0 = Roots.prepare, record s + 1 = simulation s); NULL / 0 returns to device-side sampling */
int lz_sroots_set_given(lz_roots *r, const float *h_draws, int records);
/* Roots.get_distributions ([root_num][K] visit counts) / get_sampled_actions ([root_num][K][D]) */
int lz_sroots_get_distributions(lz_roots *r, int32_t *h_out);
int lz_sroots_get_sampled_actions(lz_roots *r, float *h_out);
/* the K actions of expanded node `node` of every root, [root_num][K][D] (node 0 = the roots, node s + 1 = the node expanded by
 * simulation s): CNode::legal_actions after expand (cnode.cpp:238-327) -- observability for the exact replay gate */
int lz_sroots_get_node_actions(lz_roots *r, int node, float *h_out);


/* Gumbel MuZero -- replaces lzero/mcts/ctree/ctree_gumbel_muzero/gmz_tree.pyx (lib/cnode.cpp): roots created with
 * lz_roots_create(..., LZ_TREE_GUMBEL_MUZERO, ...); Roots.prepare / prepare_no_noise take the root rewards AND values
 * (gmz_tree.pyx:36-40); batch_traverse selects with sequential halving at the root (cselect_root_child :701-745) and the
 * completed-Q improved policy below it (cselect_interior_child :747-790); get_policies / get_children_values are
 * CRoots::get_policies / get_children_values (:506-541).  Distributions / values / trajectories: the generic getters. */
int lz_groots_prepare(lz_roots *r, float root_noise_weight, const float *h_noises_flat, const float *h_rewards,
                      const float *h_values, const float *h_policy_logits, const int32_t *h_to_play);
int lz_gbatch_traverse(lz_roots *r, int num_simulations, int max_num_considered_actions, float discount_factor,
                       int32_t *h_virtual_to_play, int32_t *h_out_index_in_search_path, int32_t *h_out_index_in_batch,
                       int32_t *h_out_last_actions, int32_t *h_out_search_lens);
int lz_gbatch_back_propagate(lz_roots *r, int current_latent_state_index, float discount_factor, const float *h_rewards,
                             const float *h_values, const float *h_policy_logits);
int lz_groots_get_policies(lz_roots *r, float discount_factor, float *h_out_policies, float *h_out_children_values);
/* GumbelMuZeroMCTSCtree.search (mcts_ctree.py:1067-1172) with an engine MuZero model, whole loop on the device; prepare the
 * roots with lz_roots_prepare_from_inference (rewards 0, values and logits of lz_initial_inference, gumbel_muzero.py:562) */
int lz_gsearch(lz_roots *r, int num_simulations, int max_num_considered_actions, float discount_factor);

/* ReZero (search_with_reuse, https://arxiv.org/abs/2404.16364) -- replaces batch_traverse_with_reuse /
 * batch_backpropagate_with_reuse (ez_tree.pyx:94-121, mz_tree.pyx:84-110; cnode.cpp:603-649, 697-754, 816-884, 965-1072).
 * h_true_action[B]: the trajectory's action at each root, h_reuse_value[B]: the value found by the search of the next state.
 * The root scores the true action with carm_score and the walk stops right below the root when it is selected;
 * h_out_index_in_search_path[i] = -1 when the node reached is already expanded (no inference for root i this simulation). */
int lz_batch_traverse_with_reuse(lz_roots *r, int pb_c_base, float pb_c_init, float discount_factor,
                                 int32_t *h_virtual_to_play, const int32_t *h_true_action, const float *h_reuse_value,
                                 int32_t *h_out_index_in_search_path, int32_t *h_out_index_in_batch,
                                 int32_t *h_out_last_actions, int32_t *h_out_search_lens);
/* value_prefixs / values / policy_logits hold n_infer rows: one per root that needed inference, in root order (the
 * reference's packed batch); h_no_inference_lst / h_reuse_lst are ascending root indices terminated by -1 (the driver's
 * lists, mcts_ctree.py:920-990); h_is_reset[B] per ROOT (NULL for the MuZero tree). */
int lz_batch_backpropagate_with_reuse(lz_roots *r, int current_latent_state_index, float discount_factor,
                                      const float *h_value_prefixs, const float *h_values, const float *h_policy_logits,
                                      int n_infer, const int32_t *h_is_reset, const int32_t *h_to_play,
                                      const int32_t *h_no_inference_lst, const int32_t *h_reuse_lst, const float *h_reuse_value);

/* the whole ReZero search on the device with an engine model (replaces EfficientZeroMCTSCtree.search_with_reuse /
 * MuZeroMCTSCtree.search_with_reuse, mcts_ctree.py:878-1002, 370-470); returns the reference's (length of the last
 * simulation's inference batch, average inference batch size). */
int lz_search_with_reuse(lz_roots *r, int num_simulations, int pb_c_base, float pb_c_init, float discount_factor,
                         int lstm_horizon_len, float value_delta_max, const int32_t *h_true_action,
                         const float *h_reuse_value, int *out_last_length, double *out_average_infer);

/* select_action (lzero/policy/utils.py:637-661) for every root on the device: p_i = N_i^(1/T) / sum over the root's legal
 * positions (float64 like the original); h_action_pos[i] = arg-max of the visit counts (deterministic != 0, np.argmax:
 * first maximum) or one draw from p (counter-based generator keyed by seed and root); h_entropy[i] in bits.
 * Also valid on Sampled-EfficientZero roots (positions = the K sampled actions). */
int lz_roots_select_action(lz_roots *r, double temperature, int deterministic, uint64_t seed, int32_t *h_action_pos,
                           double *h_entropy);

/* ------------------------------------------------------------------------------------------------
 * Network -- replaces lzero/model/efficientzero_model.py (EfficientZeroModel.initial_inference :203-238,
 * recurrent_inference :240-273) evaluated in eval() mode, with InverseScalarTransform
 * (lzero/policy/scaling_transform.py:82-92) fused into the value / value-prefix heads.
 * ---------------------------------------------------------------------------------------------- */
typedef struct lz_model_cfg {
    int model_type;         /* 0: EfficientZeroModel (conv)   1: MuZeroModel (conv)   2: MuZeroModelMLP (muzero_model_mlp.py)
                               3: EfficientZeroModelMLP (efficientzero_model_mlp.py)   4: SampledEfficientZeroModelMLP, continuous
                               actions (sampled_efficientzero_model_mlp.py) */
    int obs_c, obs_h, obs_w;/* observation_shape, e.g. 4, 96, 96; MLP models: obs_c = vector length, obs_h = obs_w = 1 */
    int action_space_size;
    int num_channels;       /* 64 (Atari, Go, Connect4), 32 (gomoku) or 16 (tictactoe) -- the narrow ones without downsample only;
                               MLP models: latent_state_dim */
    int lstm_hidden_size;   /* 512 (EfficientZero) */
    int head_channels;      /* reward/value/policy head channels (16) */
    int head_hidden;        /* hidden width of the head MLPs: 32, or less (tictactoe: 8) */
    int support_size;       /* 601 */
    float support_min;      /* -300 */
    float bn_eps;           /* 1e-5 */
    int downsample;         /* 1: DownSample tower, 96x96 obs -> 6x6 latent (Atari); 0: latent grid = obs grid (board games,
                               e.g. Go 9x9: obs 17x9x9) */
    /* ---- MLP model family (model_type >= 2; the layer widths are taken from the tensors themselves) -- and `activation` also the conv Sampled EfficientZero */
    int activation;         /* 0 ReLU, 1 GELU(approximate='tanh') */
    int res_connection_in_dynamics;
    int action_encoding;    /* 0 one_hot, 1 not_one_hot (action / action_space_size; also on the conv models: ONE action plane,
                               efficientzero_model.py:355-369), 2 continuous (the action vector) */
    int num_of_sampled_actions;  /* K (model_type 4); model_type 0 with K > 0 = SampledEfficientZeroModel (conv, sampled_efficientzero_model.py:17) with DISCRETE
                                    actions: the EfficientZero network searched by the sampled tree (roots: A = K, discrete size = action_space_size); it
                                    alone may set activation = 1 (GELU, its default) and head_hidden up to 256 (its default) on a conv model */
    int sigma_type;         /* 0 conditioned */
    int bound_type;         /* 0 None, 1 tanh on mu */
    float ln_eps;           /* 1e-5 */
    /* ---- convolutional models */
    int num_res_blocks;     /* residual blocks of the representation / dynamics / prediction networks: 1 (0 means 1), 2 or 3 */
    /* MuZeroModel (conv) only: a reward support that differs from the value support (reward_support_range, muzero_model.py; the drivers
     * build one inverse-transform handle per support, mcts_ctree.py:726-729); 0 = the value support.  The reference's EfficientZero
     * driver sends the value prefix through the VALUE handle (mcts_ctree.py:839-841), so there the two must be equal. */
    int reward_support_size;
    float reward_support_min;
    /* 0: parity mode -- fp32 arithmetic throughout (the default; every parity claim is about this mode).
     * 1: fast mode (last arm; reported separately, statistical parity only): the 3x3 convolutions of the recurrent
     *    chain and the LSTM gate product run on bf16 MFMA (weights and the multiplied activations rounded to bf16, fp32 accumulation,
     *    fp32 normalisation / cell / heads / tree).  EfficientZeroModel / MuZeroModel (conv) on 4x96x96 (6x6x64 latent) or 4x64x64 (8x8x64 latent) observations only.
     * 2: parity mode on the fp32 matrix instructions only: the 3x3 convolutions of the tower and of the recurrent chain keep the kernels
     *    of rounds 1-4 (Winograd / direct form on v_mfma_f32_*) instead of the split-bf16 products (three exact bf16 planes per operand, six
     *    plane products per k-step) that mode 0 uses since round 5 -- the same 1e-5 (1 + |x|) parity bound, other roundings, slower; per
     *    model what LZ_CHAIN_NO_SPLIT=1 LZ_CONV_NO_SPLIT=1 select per process. */
    int precision;
    /* ---- MLP model family, two more keywords of the reference constructors (muzero_model_mlp.py:30,33; efficientzero_model_mlp.py:32,34;
     * sampled_efficientzero_model_mlp.py:33,36) */
    int state_norm;         /* 1: state_norm=True -- the latent state is renormalised to [0, 1] over its features after the representation and
                               after the dynamics network (lzero/model/utils.py:242-271: (x - min) / max(max - min, 1e-8)); the reward / value-prefix
                               path reads the un-normalised next latent, as in the reference's dynamics networks */
    int scalar_heads;       /* 1: categorical_distribution=False -- value and reward / value-prefix heads have ONE output, the scaled scalar itself:
                               h^-1 is applied to it directly (scaling_transform.py:84-92); support_size (and reward_support_size) must be 1 */
} lz_model_cfg;

/* One model per engine (creating another replaces it: roots of the old one re-size their pools on the next inference).
 * Calling lz_model_set_tensor + lz_model_finalize again on a live engine is a WEIGHT REFRESH (collector after a learner
 * update): tensors of unchanged shape are overwritten in place, captured search graphs stay valid.  Weights are ingested by their reference state_dict names
 * (e.g. "dynamics_network.lstm.weight_ih_l0", "prediction_network.fc_value.3.bias"), fp32, C-contiguous,
 * host memory; lz_model_finalize folds the eval-mode BatchNorms, re-lays the tensors out for the
 * kernels and uploads them.  Replaces policy._collect_model.load_state_dict (muzero.py:1036-1058 format). */
int lz_model_create(lz_engine *e, const lz_model_cfg *cfg);
int lz_model_set_tensor(lz_engine *e, const char *name, const float *h_data, const int64_t *shape, int ndim);
/* the same from a device tensor (weight refresh after an RCCL broadcast: no host round trip in the caller; the library copies it to its
 * host-side staging on the engine's stream -- the re-layout of lz_model_finalize is host code).  The producing stream must be done with d_data. */
int lz_model_set_tensor_device(lz_engine *e, const char *name, const float *d_data, const int64_t *shape, int ndim);
int lz_model_finalize(lz_engine *e);
/* Weight REFRESH of a loaded convolutional fp32 model without the host (a collector after a learner update; in the reference the collector
 * plays with the learner's own nn.Module, lzero/policy/muzero.py:1049-1061, lzero/entry/train_muzero.py:187-212 -- fresh weights are free there).
 * lz_model_flat_layout / _entry: the model's state_dict tensors (no num_batches_tracked) in name order with their offsets in ONE flat fp32
 * buffer.  lz_model_refresh_flat: that buffer (device pointer when on_device != 0, else host) -> every kernel layout (MFMA-fragment orders,
 * Winograd U = G g G^T in binary64, folded BatchNorm, action table, ...) by kernels on the engine's stream; bit-identical to
 * lz_model_set_tensor + lz_model_finalize of the same tensors; same device buffers (captured search graphs stay valid); no host
 * synchronisation.  LZ_ERR_STATE (with the reason) where no device-side refresh exists (MLP models, fast mode): use the calls above. */
int lz_model_flat_layout(lz_engine *e, int64_t *out_tensors, int64_t *out_floats);
int lz_model_flat_entry(lz_engine *e, int64_t i, char *name_buf, int64_t name_buf_len, int64_t *out_offset, int64_t *out_size);
int lz_model_flat_host_buffer(lz_engine *e, float **out);   /* pinned staging for a host-side flat state_dict; passing it to lz_model_refresh_flat skips a copy */
int lz_model_refresh_flat(lz_engine *e, const float *flat, int64_t n_floats, int on_device);
/* FNV-1a over every device weight buffer (parity tests of the refresh path) */
int lz_model_weights_digest(lz_engine *e, uint64_t *out);
/* Identity of the model an engine currently holds: incremented by every lz_model_create.  A host-side model object records
 * it and refuses to run once another model took its engine (one model per engine). */
uint64_t lz_engine_model_uid(lz_engine *e);

/* initial_inference for the roots' batch: d_obs is NCHW fp32 [root_num][obs_c][obs_h][obs_w] in HBM.
 * The latent state goes to slot 0 of the roots' latent pool, LSTM state slot 0 is zeroed
 * (efficientzero_model.py:229-238).  Predicted values (after h^-1) and policy logits stay in HBM and
 * can be fetched with lz_roots_get_root_outputs. */
int lz_initial_inference(lz_roots *r, const float *d_obs);
/* same from a HOST observation batch (staged through HBM; the PCIe copy is on the engine stream) */
int lz_initial_inference_host(lz_roots *r, const float *h_obs);
int lz_roots_get_root_outputs(lz_roots *r, float *h_pred_values, float *h_policy_logits);
/* The reference's call order infers before the roots exist: network_output = model.initial_inference(obs); roots = MCTSCtree.roots(n,
 * legal); roots.prepare(noise_w, noises, value_prefix_roots, policy_logits, to_play); search(roots, model, latent_state_roots,
 * reward_hidden_state_roots, to_play) (lzero/policy/efficientzero.py:582-610).  The engine model infers into a handle of its own
 * (src); this call moves that inference (slot 0 of the latent / LSTM pools, root predictions) device-to-device into the roots the
 * caller prepared from host lists (dst, same engine / root_num / action space), after which lz_search(dst, ...) runs the fused loop. */
int lz_roots_adopt_inference(lz_roots *dst, lz_roots *src);
/* everything _forward_collect reads back after a fused search (efficientzero.py:616-643) in one readout launch, one
 * device-to-host copy and one synchronisation: get_distributions (+ counts), get_values, and the root predictions of
 * lz_initial_inference (h_pred_values / h_policy_logits may be NULL) */
int lz_roots_get_search_results(lz_roots *r, int32_t *h_out_dist, int32_t *h_out_count, float *h_out_values,
                                float *h_pred_values, float *h_policy_logits);
/* The same read-back with lz_roots_select_action (select_action, lzero/policy/utils.py:637-661) for every root computed on
 * the device before it, so that a collect forward has ONE device-host synchronisation after the search: h_action_pos
 * [root_num] positions in the legal list, h_entropy [root_num] bits. */
int lz_roots_get_search_results_select(lz_roots *r, int32_t *h_out_dist, int32_t *h_out_count, float *h_out_values,
                                       float *h_pred_values, float *h_policy_logits, double temperature, int deterministic,
                                       uint64_t seed, int32_t *h_action_pos, double *h_entropy);
/* Env-step rows -- the collector-side glue: everything MuZeroCollector keeps per env and step
 * (muzero_collector.py:557-620) in the field set of GameSegment.append / store_search_stats (game_segment.py:158-182, 241-263),
 * written by one kernel from device buffers after a fused search, select_action (lzero/policy/utils.py:637-661) included.
 * Row = row_words float32 (>= lz_rows_width(A, frame_floats)):
 *   [0] action  [1] reward (0: filled by the environment side)  [2] searched root value  [3] predicted value  [4] to_play
 *   [5] timestep (h_timestep[i], -1 when NULL)  [6] visit-count entropy (bits)  [7] number of legal actions
 *   [8, 8+A) child visits / sum in legal-list order (game_segment.py:247-252)   [8+A, 8+2A) action mask
 *   [8+2A, 8+2A+frame_floats) the newest observation frame = the LAST frame_floats floats of each env's observation
 *   (d_obs; NULL = the batch of the latest lz_initial_inference, whose buffer the caller must then keep alive and unchanged until
 *   this call returns -- pass the pointer explicitly when an allocator may have recycled it).
 * d_rows: DEVICE [root_num][row_words] -- the payload of the trajectory all-gather (lightzero_amd/shard.py), it never
 * crosses PCIe here.  h_header [root_num][8+2A]: host copy of the row headers (what stepping the environments needs);
 * h_policy_logits [root_num][A] may be NULL.  A <= 64: ONE launch (readout + select_action + packing; the header words and the
 * logits are written by the kernel into the library's pinned host block) and one synchronisation; A > 64: readout, select_action and
 * pack launches, two copies, one synchronisation. */
int lz_rows_width(int action_space_size, int frame_floats);
/* Host only: the Winograd F(2x2, 3x3) weight transform U = G g G^T the convolution kernels ingest (computed in binary64, rounded once),
 * in a plain layout: w [cout][cin][3][3] (the reference's Conv2d weight, e.g. common.py:309-327 ResBlock convolutions) ->
 * u [16 points][cin][cout].  Exists so that a CPU test can pin the transform against an independent statement of the algorithm. */
int lz_wino_weights(const float *w, int cout, int cin, float *u);
int lz_roots_collect_rows(lz_roots *r, double temperature, int deterministic, uint64_t seed, const float *d_obs,
                          int frame_floats, const int32_t *h_timestep, float *d_rows, int row_words, float *h_header,
                          float *h_policy_logits);
/* The rows of the two further families (game_segment.py:254-258, muzero_collector.py:606-612): an EXTRA block of lz_rows_extra_words(r)
 * floats between the action mask and the frame --
 *   Sampled EfficientZero roots: root_sampled_actions [K][D]; visit block / mask / n_legal are over the K sampled actions, word 0 is
 *     the selected POSITION (its action = extra[pos * D, pos * D + D));
 *   Gumbel MuZero roots: improved_policy_probs [A] (CRoots::get_policies, cnode.cpp:506-541, with discount_factor); word 0 = arg-max of
 *     the improved policy over the legal actions (lzero/policy/gumbel_muzero.py:591-592);
 *   EfficientZero / MuZero roots: no extra block (the call is lz_roots_collect_rows).
 * row_words >= 8 + 2 A + extra + frame_floats; h_header is [root_num][8 + 2 A + extra]. */
int lz_rows_extra_words(lz_roots *r);
int lz_roots_collect_rows_ex(lz_roots *r, double temperature, int deterministic, uint64_t seed, float discount_factor, const float *d_obs,
                             int frame_floats, const int32_t *h_timestep, float *d_rows, int row_words, float *h_header,
                             float *h_policy_logits);
/* The same in two halves: _begin enqueues select_action + the row packing behind the search and returns at once; _end waits for THOSE
 * rows only (an event, not the stream: the search of another roots handle may already be queued behind them on the same engine -- the
 * vectorised collector keeps the device busy with one env group while the host steps the other's environments,
 * muzero_collector.py:557-692 is a strictly serial loop) and copies the header words (and the root policy logits when they were asked
 * for) out.  One pair in flight per roots handle. */
int lz_roots_collect_rows_begin(lz_roots *r, double temperature, int deterministic, uint64_t seed, float discount_factor, const float *d_obs,
                                int frame_floats, const int32_t *h_timestep, float *d_rows, int row_words, int want_logits);
int lz_roots_collect_rows_end(lz_roots *r, float *h_header, float *h_policy_logits);
/* Roots.prepare / prepare_no_noise with the policy logits of lz_initial_inference (value prefix 0 for
 * EfficientZero, efficientzero_model.py:238).  h_noises_flat as in lz_roots_prepare (NULL: no noise). */
int lz_roots_prepare_from_inference(lz_roots *r, float root_noise_weight, const float *h_noises_flat,
                                    const int32_t *h_to_play);
/* The same prepare with the exploration noise DRAWN ON THE DEVICE: Dirichlet(root_dirichlet_alpha) over every root's legal actions
 * (lzero/policy/efficientzero.py:599-602 draws np.random.dirichlet([alpha] * n_legal) per env on the host), gamma variates by
 * Marsaglia-Tsang from a counter-based generator keyed by (roots seed, device-resident epoch that every prepare advances, root,
 * legal position); only to_play crosses PCIe.  Distribution-equivalent to the reference, not stream-equivalent (parity runs inject
 * their noise through lz_roots_prepare_from_inference). */
int lz_roots_prepare_from_inference_dirichlet(lz_roots *r, float root_noise_weight, float root_dirichlet_alpha,
                                              const int32_t *h_to_play);
/* EfficientZeroMCTSCtree.search (lzero/mcts/tree_search/mcts_ctree.py:745-876): num_simulations x
 * [select -> gather latent/LSTM state by (ix, iy) -> recurrent_inference -> h^-1 -> LSTM reset every
 * lstm_horizon_len -> expand + backup], entirely on the device, no host synchronisation inside.
 * Asynchronous on the engine stream; results via lz_roots_get_distributions / get_values. */
int lz_search(lz_roots *r, int num_simulations, int pb_c_base, float pb_c_init, float discount_factor,
              int lstm_horizon_len, float value_delta_max);

/* observability for parity tests: per-simulation records and pools (host copies)                  */
/* tracing is off by default; when on, the captured search graph carries one extra D2D copy per simulation and the head
 * kernels also write their support-wide logits (lz_roots_read_debug_logits); on & 2: head debug buffers too (lz_roots_read_head_debug) */
int lz_roots_enable_trace(lz_roots *r, int on);
int lz_roots_read_trace(lz_roots *r, int num_simulations, int32_t *h_out /* [S][B][4] ix, action, search_len, to_play */);
/* depth in its tree of the node every simulation expanded == that simulation's search-path length (CSearchResults::search_lens,
 * cnode.h:66-80, which the reference keeps for the latest simulation only): h_out [root_num][num_simulations], node s + 1 of root i
 * at h_out[i * num_simulations + s].  Valid after a search of the EfficientZero / MuZero tree (not the sampled / Gumbel trees);
 * no tracing needed -- the depths are part of the node records (lz_tree_dev::node_link).  bench.py reports their mean / max. */
int lz_roots_get_node_depths(lz_roots *r, int num_simulations, int32_t *h_out);
int lz_roots_read_sim_outputs(lz_roots *r, int slot, float *h_value_prefix, float *h_value, float *h_policy_logits);
int lz_roots_read_latent(lz_roots *r, int slot, float *h_out_nchw);
int lz_roots_read_hidden(lz_roots *r, int slot, float *h_h, float *h_c);
/* teacher-forced / foreign-driver access: put a caller-provided latent state (NCHW [root_num][C][H][W]; MLP models [root_num][L])
 * and LSTM state ([root_num][H] each) into pool slot `slot` */
int lz_roots_write_latent(lz_roots *r, int slot, const float *h_in_nchw);
int lz_roots_write_hidden(lz_roots *r, int slot, const float *h_h, const float *h_c);
/* Model.recurrent_inference(latent_state, reward_hidden_state, action)   efficientzero_model.py:240-273, muzero_model.py:240-272
 * on pool slots: root i reads the state of slot h_parent_slot[i] and takes h_actions[i] (h_actions_f [root_num][D] for
 * Sampled-EfficientZero roots); next latent / LSTM state / value prefix|reward / value / policy logits land in out_slot
 * (lz_roots_read_latent / _hidden / _sim_outputs).  h_search_len (may be NULL) and lstm_horizon_len reproduce the driver's LSTM
 * reset rule search_len % lstm_horizon_len == 0 (mcts_ctree.py:859-863). */
int lz_recurrent_inference(lz_roots *r, const int32_t *h_parent_slot, const int32_t *h_actions, const float *h_actions_f,
                           const int32_t *h_search_len, int lstm_horizon_len, int out_slot);
/* optional debug logits of the last head launch: which = 0 value [B][support], 1 value_prefix/reward [B][support] */
int lz_roots_read_debug_logits(lz_roots *r, int which, float *h_out);
/* in-stream timing (HIP events on the engine stream) of the 64->64 3x3 convolution on the latent grid --
 * the dominant kernel of the recurrent loop -- for bench.py's roofline object */
int lz_profile_enable(lz_engine *e, int max_launches);
int lz_profile_read(lz_engine *e, int64_t *out_launches, double *out_total_ms);
/* In-graph timing of the two launches of a simulation (bench.py's roofline clock of THIS run; HIP events cannot be recorded inside a
 * captured graph): while on, the first workgroup of every chain and LSTM launch stores its s_memrealtime start and the last one (by
 * block id) its end (100 MHz constant-rate counter, 10 ns ticks) into per-simulation words; the search graph is re-captured with the stamp pointers.
 * lz_roots_read_stamps: h_out [num_simulations][4] = {chain: first workgroup's start, last workgroup's end, LSTM: the same}. */
int lz_roots_enable_stamps(lz_roots *r, int on);
int lz_roots_read_stamps(lz_roots *r, int num_simulations, uint64_t *h_out);
/* lz_roots_enable_trace(r, 3): tracing + head debug buffers -- the support-wide logits [B][support] and the pre-transform expectation
 * softmax . support [B] of the value (which = 0) / value-prefix | reward (which = 1) head at pool slot `slot`, for EVERY simulation and
 * whichever kernel finished the head (the head launch, or the split heads inside the next chain launch).  Replaces nothing in the
 * reference: InverseScalarTransform.__call__ (scaling_transform.py:82-92) computes both inside one expression; the parity tests
 * compare them with torch at north_star's 1e-5. */
int lz_roots_read_head_debug(lz_roots *r, int slot, int which, float *h_logits, float *h_expect);
/* h_out[i] = the device's inverse scalar transform of h_in[i] (scaling_transform.py:88-91 in fp32, torch's operation order), evaluated
 * by the copy compiled into the conv-model head kernels (which = 0) or the MLP family's row finisher (which = 1): the parity test holds
 * both BIT-EQUAL to torch on >= 10^6 inputs over the support range. */
int lz_debug_inverse_scalar_transform(lz_engine *e, int which, const float *h_in, int64_t n, float *h_out);
/* debugging aids: stop lz_initial_inference after stage k ("stop_stage"), read a workspace buffer */
int lz_debug_set(lz_engine *e, const char *key, int value);
int lz_debug_read_ws(lz_engine *e, int which, float *h_out, int64_t n);
int lz_debug_read_param(lz_engine *e, const char *name, float *h_out, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* LZ_MI355_H */
