// lz_mlp.h -- vector-observation (MLP) model family, see lz_mlp.hip
#pragma once
#include "lz_internal.h"

struct lz_mlp_model;
struct lz_model;
void lz_mlp_model_destroy(lz_mlp_model *mm);
int lz_mlp_finalize(lz_engine *e);
int lz_mlp_ensure_pools(lz_roots *r);
int lz_mlp_initial_inference(lz_roots *r, const float *d_obs);
void lz_mlp_recurrent(lz_roots *r, int sim, int horizon, hipStream_t s);
int lz_mlp_latent_size(const lz_model *m);
int lz_mlp_hidden_size(const lz_model *m);
int lz_mlp_policy_width(const lz_model *m);
int lz_mlp_obs_size(const lz_model *m);
