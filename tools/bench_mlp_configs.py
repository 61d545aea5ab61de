#!/usr/bin/env python
"""Timing of the vector-observation configurations of BASELINE.json on one MI355X (not the headline bench line):
configs[0] CartPole MuZero MLP (8 envs x 25 sims) and configs[4] Sampled EfficientZero DMC state (K = 20, 64 envs per GPU
x 50 sims; --envs 256 for the whole 4-GPU batch on one device).  Search only (initial inference -> prepare -> fused
search -> read-back), inputs resident in HBM, synthetic seeded weights in the reference's state_dict layout
(lightzero_amd.model.synthetic.mlp_state_dict; tensor shapes committed in lightzero_amd/model/synthetic_mlp_specs.json).

    python tools/bench_mlp_configs.py --config 4 --envs 256 --steps 50
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=4, choices=[0, 4])
    ap.add_argument("--envs", type=int, default=None)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    args = ap.parse_args()
    import torch
    from lightzero_amd.model.synthetic import mlp_state_dict
    from lightzero_amd import _lib as L
    lib = L.lib()
    if args.config == 0:
        from lightzero_amd.model.muzero_model_mlp import MuZeroModelMLP
        from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree
        B, S, A, OBS = args.envs or 8, 25, 2, 4
        model = MuZeroModelMLP(observation_shape=OBS, action_space_size=A, latent_state_dim=128).load_state_dict(mlp_state_dict("muzero_mlp_cartpole"))
        roots = mz_tree.Roots(B, [list(range(A))] * B, action_space_size=A, max_simulations=S)
        roots.set_tiebreak(1, seed=1)
        roots._ensure(A)
        horizon, name = 0, "configs[0] CartPole MuZeroModelMLP"
    else:
        from lightzero_amd.model.sampled_efficientzero_model_mlp import SampledEfficientZeroModelMLP
        from lightzero_amd.mcts.ctree.ctree_sampled_efficientzero import ezs_tree
        B, S, A, OBS, K = args.envs or 64, 50, 1, 5, 20
        model = SampledEfficientZeroModelMLP(observation_shape=OBS, action_space_size=A, continuous_action_space=True,
                                             num_of_sampled_actions=K).load_state_dict(mlp_state_dict("sampled_efficientzero_mlp_dmc"))
        roots = ezs_tree.Roots(B, [[-1] * K] * B, A, K, True, max_simulations=S)
        horizon, name = 5, "configs[4] DMC-state SampledEfficientZeroModelMLP K=20"
    obs = torch.rand(B, OBS, generator=torch.Generator().manual_seed(0)).cuda().contiguous()
    to_play = L.i32([-1] * B)
    rng = np.random.default_rng(0)
    width = roots.K if args.config == 4 else A
    dist = np.zeros((B, width), np.int32); cnt = np.zeros(B, np.int32); val = np.zeros(B, np.float32)

    def step():
        L.check(lib.lz_initial_inference(roots._h, obs.data_ptr()))
        if args.config == 0:
            nz = rng.dirichlet([0.3] * A, size=B).astype(np.float32)
            L.check(lib.lz_roots_prepare_from_inference(roots._h, 0.25, nz.ctypes.data, to_play))
        else:
            L.check(lib.lz_roots_prepare_from_inference(roots._h, 0.25, None, to_play))
        L.check(lib.lz_search(roots._h, S, 19652, 1.25, 0.997, horizon, 0.01))
        if args.config == 0:
            L.check(lib.lz_roots_get_distributions(roots._h, dist, cnt))
        else:
            L.check(lib.lz_sroots_get_distributions(roots._h, dist))
        L.check(lib.lz_roots_get_values(roots._h, val))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"workload": name, "envs": B, "num_simulations": S, "ms_per_step": dt * 1e3, "env_steps_per_s": B / dt,
                      "mcts_sims_per_s": B * S / dt}))


if __name__ == "__main__":
    main()
