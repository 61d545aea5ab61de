"""Search-only timing of the conv-model configurations on one MI355X.  Default: BASELINE.json configs[2], Atari MuZero (obs
4x96x96, A = 4), 1024 envs x 400 simulations -- the deep-tree stress case.  obs already in HBM -> distributions / root
values on the host.  Synthetic seed-0 weights (lightzero_amd.model.synthetic), random obs.

    python tools/bench_conv_configs.py [--envs 1024] [--sims 400] [--steps 5] [--warmup 2] [--streams 2]
    python tools/bench_conv_configs.py --family ez --obs 64 --envs 256 --sims 50 --actions 6 --steps 20
        (the reference's shipped Atari EfficientZero configuration: 64x64 observations, 8x8 latent, support (-50, 51, 1))
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--actions", type=int, default=4)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--family", choices=["mz", "ez"], default="mz")
    ap.add_argument("--obs", type=int, choices=[96, 64], default=96)
    ap.add_argument("--go", action="store_true", help="BASELINE configs[3] per-GPU share instead: Go 9x9 MuZero (obs 17x9x9, no downsample, "
                    "A = 82, two players); use with --envs 64 --sims 200")
    ap.add_argument("--fast", action="store_true", help="fast mode (bf16 matrix products; 4x96x96 observations): a separate arm, statistical parity only")
    ap.add_argument("--streams", type=int, default=1, help="independent sub-batches, each on its own engine / HIP stream: the "
                    "latency-bound tree step of one overlaps the MFMA-bound network step of another")
    ap.add_argument("--chain-ts", action="store_true", help="debug build only (LZ_MI355_LIB=lightzero_amd/liblz_mi355_dbg.so LZ_DEBUG_CHAIN_TS=1 "
                    "LZ_NO_GRAPH=1): print the cycle stamps of the last chain launch's workgroup 0 (k_chain_s3g)")
    a = ap.parse_args()
    import numpy as np
    import torch
    from lightzero_amd import _lib as L
    from lightzero_amd.model.synthetic import efficientzero_state_dict
    if a.family == "mz":
        from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree as tree
        from lightzero_amd.model.muzero_model import MuZeroModel as Model
    else:
        from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree as tree
        from lightzero_amd.model.efficientzero_model import EfficientZeroModel as Model
    import ctypes
    B, A, S, NS = a.envs, a.actions, a.sims, max(1, a.streams)
    assert B % NS == 0
    EPS = B // NS
    sup = (-300., 301., 1.) if a.obs == 96 else (-50., 51., 1.)
    if a.go:
        A = 82
        weights = efficientzero_state_dict(seed=0, action_space_size=A, muzero=a.family == "mz", latent_pixels=81, observation_channels=17,
                                           downsample=False)
        mkw = dict(observation_shape=(17, 9, 9), action_space_size=A, downsample=False)
        obs = (torch.rand(B, 17, 9, 9, generator=torch.Generator().manual_seed(1)) < 0.3).float().cuda().contiguous()
    else:
        weights = efficientzero_state_dict(seed=0, action_space_size=A, muzero=a.family == "mz", latent_pixels=36 if a.obs == 96 else 64,
                                           support_size=int(sup[1] - sup[0]))
        mkw = dict(observation_shape=(4, a.obs, a.obs), action_space_size=A, reward_support_range=sup, value_support_range=sup)
        if a.fast:
            mkw["fast_mode"] = True
        obs = torch.rand(B, 4, a.obs, a.obs, generator=torch.Generator().manual_seed(1)).cuda().contiguous()
    torch.cuda.synchronize()
    legal = [list(range(A))] * EPS
    to_play = ([1, 2] * EPS)[:EPS] if a.go else [-1] * EPS
    rng = np.random.default_rng(0)
    noises = rng.dirichlet([0.3] * A, size=B).astype(np.float32)
    parts = []
    for k in range(NS):
        if k == 0:
            e = L.default_engine(0)
        else:
            e = L.P()
            L.check(L.lib().lz_engine_create(0, ctypes.byref(e)))
        model = Model(engine=e, **mkw).load_state_dict(weights)
        roots = tree.Roots(EPS, legal, action_space_size=A, max_simulations=S, engine=e)
        parts.append((model, roots, obs[k * EPS:(k + 1) * EPS].contiguous(), np.ascontiguousarray(noises[k * EPS:(k + 1) * EPS])))

    def step():
        for model, roots, o, nz in parts:  # enqueue every sub-batch before reading anything back
            roots.reset(legal)
            model.initial_inference(o, roots, fetch=False)
            roots.prepare_from_inference(0.25, nz, to_play)
            L.check(L.lib().lz_search(roots._h, S, 19652, 1.25, 1.0 if a.go else 0.997, 5 if a.family == "ez" else 0, 0.01))
        res = [roots.get_search_results() for _, roots, _, _ in parts]
        return [np.concatenate([r[i] for r in res]) for i in range(5)]

    for _ in range(a.warmup):
        res = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        res = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    assert (np.asarray(res[0]).sum(1) == S).all()
    if a.chain_ts:
        L.lib().lz_debug_read_chain_ts.argtypes = [ctypes.c_void_p]
        out = np.zeros(64, np.uint64)
        L.check(L.lib().lz_debug_read_chain_ts(out.ctypes.data))
        n = int(out[0]); ts = out[1:1 + n].astype(np.int64)
        names = ["start", "staged", "sync"] + [x for l in range(14) for x in ("L%d products" % l, "L%d exchange barrier" % l, "L%d epilogue" % l, "L%d barrier" % l)]
        for i in range(1, n):
            print("%-22s +%7d cycles   (t=%7d)" % (names[i] if i < n - 1 else "end (1x1 convs)", ts[i] - ts[i - 1], ts[i] - ts[0]), file=sys.stderr)
        w0, w1 = out[32:40].astype(np.int64) - int(ts[0]), out[40:48].astype(np.int64) - int(ts[0])
        ex = out[50:59].astype(np.int64) - int(ts[0])
        print("prologue: selection known %d, latent requested %d, latent in LDS %d, halo/zero %d, BN tables %d, 1x1 params %d | 1x1 unit 0: operands %d, MFMAs done %d, stored %d (end of layers %d)" % (tuple(ex.tolist()) + (int(ts[n - 2] - ts[0]),)), file=sys.stderr)
        print("layer 2 products per wave: start", w0.tolist(), "end", w1.tolist(), file=sys.stderr)
    print(json.dumps({"workload": "Go 9x9 MuZero conv (configs[3] share)" if a.go else "Atari %s conv, obs %dx%d" % ("MuZero" if a.family == "mz" else "EfficientZero", a.obs, a.obs), "envs": B, "num_simulations": S, "actions": A, "sub_batches": NS, "mode": "fast (bf16 products)" if a.fast else "parity (fp32)",
                      "ms_per_step": dt * 1e3, "env_steps_per_s": B / dt, "mcts_sims_per_s": B * S / dt}))


if __name__ == "__main__":
    main()
