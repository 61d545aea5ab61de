#!/usr/bin/env python
"""Headline benchmark: self-play env-steps/s (and MCTS sims/s) for EfficientZero Atari, 96x96x4
observations, 50 simulations, 256 parallel envs per MI355X (BASELINE.json configs[1]).

A "step" is one pass of the hot path over one batch of 256 synthetic observations that are already
resident in HBM: initial_inference -> root prepare with Dirichlet noise -> 50 x [select -> recurrent
inference -> expand/backup] on the device -> visit-count distributions and root values on the host
(the `_forward_collect` contract).  Weak scaling: every GPU owns its own 256 envs; the only collective
is the all-gather of the packed trajectory rows (lightzero_amd/shard.py).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS, SIMS, ACTIONS = 256, 50, 6
FLOP_RECURRENT = 18381312        # per env per simulation
FLOP_INITIAL = 292222912         # per env-step
FLOP_CHAIN = 2 * 36 * 64 * (70 + 4 * 64) * 9 + 2 * 36 * 64 * 48  # dyn conv 70->64 + 4 convs 64->64 + three 1x1 64->16 = 13,741,056 per env
PEAK_FP32_MATRIX_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32 dense peak
CFG = dict(num_simulations=SIMS, pb_c_base=19652, pb_c_init=1.25, discount_factor=0.997, value_delta_max=0.01,
           lstm_horizon_len=5, root_noise_weight=0.25, root_dirichlet_alpha=0.3)


def _reference_model(weights):
    """the oracle's torch restatement of EfficientZeroModel carrying the benchmark's weights (baseline legs only)"""
    import torch
    from oracle import torch_models as tm
    m = tm.EfficientZeroModel(action_space_size=ACTIONS)
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()}, strict=False)
    assert not missing.unexpected_keys and all(k.endswith("num_batches_tracked") for k in missing.missing_keys), missing
    return m.eval()


def cpu_baseline(weights, obs_cpu, noises, budget_s=25.0, device="cpu"):
    """The reference pipeline on the host cores: reference ctree (compiled from its own sources when
    oracle/_ref is present, else the C restatement) + restated Python driver + torch fp32 model.
    device="cuda" is SURVEY.md 8d's second arm, the reference as deployed with cuda=True: same driver and
    host ctree, torch model on the MI355X through stock PyTorch-ROCm (MIOpen / rocBLAS)."""
    import torch
    ref_model = _reference_model(weights)
    if device != "cpu":
        ref_model = ref_model.to(device)
        obs_cpu = obs_cpu.to(device)
    from oracle import build_ref, ctree as octree, search as osearch
    mods = build_ref.load("stock")
    kind_tree = "reference ctree (oracle/_ref/stock)" if mods else "C restatement of the ctree (oracle/ctree_oracle.c)"
    tree = mods[0] if mods else octree.ez_tree
    kw = {} if mods else dict(roots_kwargs=dict(action_space_size=ACTIONS, max_simulations=SIMS))
    cores = torch.get_num_threads()
    legal = [list(range(ACTIONS))] * ENVS
    n, t_used = 0, 0.0
    kw["device"] = device
    osearch.ez_forward_collect(tree, ref_model, obs_cpu[:32], legal[:32], [z for z in noises[:32]], [-1] * 32, CFG, **kw)  # warm-up
    if device != "cpu":
        osearch.ez_forward_collect(tree, ref_model, obs_cpu, legal, noises, [-1] * ENVS, CFG, **kw)  # MIOpen solver search at B=256
    while t_used < budget_s * 0.5 and n < 3:
        t0 = time.perf_counter()
        osearch.ez_forward_collect(tree, ref_model, obs_cpu, legal, noises, [-1] * ENVS, CFG, **kw)
        t_used += time.perf_counter() - t0
        n += 1
    if device != "cpu":
        return dict(value=ENVS * n / t_used, unit="env-steps/s", cores=1, kind="port",
                    sample="%d full env-step batches after warm-up; %s on 1 host thread + restated driver + torch fp32 model on "
                           "the MI355X via stock PyTorch-ROCm (the reference with cuda=True); %.1f s" % (n, kind_tree, t_used))
    return dict(value=ENVS * n / t_used, unit="env-steps/s", cores=cores, kind="port",
                sample="%d full env-step batches (256 envs x 50 sims) after a 32-env warm-up; %s + restated "
                       "EfficientZeroMCTSCtree.search driver + torch fp32 model on %d threads; %.1f s" %
                       (n, kind_tree, cores, t_used))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=1,
                    help="split the 256 envs of a GPU into this many independent sub-batches, each on its own engine "
                         "(HIP stream): the MFMA-bound conv chain of one overlaps the latency-bound tree / LSTM / head "
                         "kernels of the other")
    ap.add_argument("--tiebreak", choices=["first", "random"], default="random",
                    help="random = the reference's stochastic tie rule (default, like collection); first = parity mode")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from lightzero_amd import _lib as L, shard
    from lightzero_amd.model.synthetic import efficientzero_state_dict  # seeded random-init weights (no checkpoints offline)
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    lib = L.lib()
    weights = efficientzero_state_dict(seed=0, action_space_size=ACTIONS)
    NS = max(1, args.streams)
    assert ENVS % NS == 0
    EPS = ENVS // NS  # envs per sub-batch
    engs, models = [], []
    for k in range(NS):
        if k == 0:
            e = L.default_engine(local_rank)
        else:
            e = L.P()
            L.check(lib.lz_engine_create(local_rank, ctypes.byref(e)))
        engs.append(e)
        models.append(EfficientZeroModel(action_space_size=ACTIONS, engine=e).load_state_dict(weights))
    eng = engs[0]
    g = torch.Generator().manual_seed(1000 + rank)
    obs_cpu = torch.rand(ENVS, 4, 96, 96, generator=g)
    obs = obs_cpu.cuda().contiguous()
    rng = np.random.default_rng(rank)
    total = args.warmup + args.steps
    noise_steps = [rng.dirichlet([0.3] * ACTIONS, size=ENVS).astype(np.float32) for _ in range(total)]
    legal = [list(range(ACTIONS))] * EPS
    roots_l = []
    for k in range(NS):
        r = ez_tree.Roots(EPS, legal, action_space_size=ACTIONS, max_simulations=SIMS, engine=engs[k])
        r.set_tiebreak(0 if args.tiebreak == "first" else 1, seed=rank * 16 + k + 1)
        r._ensure(ACTIONS)
        roots_l.append(r)
    to_play = L.i32([-1] * EPS)
    obs_parts = [obs[k * EPS:(k + 1) * EPS].contiguous() for k in range(NS)]
    dist_out = np.zeros((ENVS, ACTIONS), np.int32)
    cnt_out = np.zeros(ENVS, np.int32)
    val_out = np.zeros(ENVS, np.float32)
    pred_out = np.zeros(ENVS, np.float32)
    logit_out = np.zeros((ENVS, ACTIONS), np.float32)
    rows_dev = torch.zeros(ENVS, 4 + ACTIONS, device="cuda")

    def step(i):
        for k, r in enumerate(roots_l):  # enqueue everything of every sub-batch before reading anything back
            L.check(lib.lz_initial_inference(r._h, obs_parts[k].data_ptr()))
            nz = np.ascontiguousarray(noise_steps[i][k * EPS:(k + 1) * EPS])
            L.check(lib.lz_roots_prepare_from_inference(r._h, CFG["root_noise_weight"], nz.ctypes.data, to_play))
            L.check(lib.lz_search(r._h, SIMS, CFG["pb_c_base"], CFG["pb_c_init"], CFG["discount_factor"],
                                  CFG["lstm_horizon_len"], CFG["value_delta_max"]))
        for k, r in enumerate(roots_l):
            sl = slice(k * EPS, (k + 1) * EPS)
            d = np.zeros((EPS, ACTIONS), np.int32); c = np.zeros(EPS, np.int32); v = np.zeros(EPS, np.float32)
            p = np.zeros(EPS, np.float32); lg = np.zeros((EPS, ACTIONS), np.float32)
            L.check(lib.lz_roots_get_search_results(r._h, d, c, v, p.ctypes.data, lg.ctypes.data))
            dist_out[sl], cnt_out[sl], val_out[sl], pred_out[sl], logit_out[sl] = d, c, v, p, lg
        if world > 1:  # pool the finished env-step rows of all ranks (RCCL all-gather over xGMI)
            rows = np.concatenate([np.zeros((ENVS, 1), np.float32), val_out[:, None], pred_out[:, None],
                                   cnt_out[:, None].astype(np.float32), dist_out.astype(np.float32)], 1)
            rows_dev.copy_(torch.from_numpy(rows))
            shard.all_gather_rows(rows_dev)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    for e in engs:
        L.check(lib.lz_engine_synchronize(e))
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        step(i)
    for e in engs:
        L.check(lib.lz_engine_synchronize(e))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert (dist_out.sum(1) == SIMS).all(), "search did not run all simulations"
    # Roofline pass: the timed region replays the search from a captured HIP graph, which cannot carry event
    # records, so the dominant kernel is timed right after it, same process and inputs, with HIP event pairs recorded
    # on the engine stream around every k_chain launch of `prof_steps` eagerly launched steps.
    prof_steps = min(args.steps, 5)
    L.check(lib.lz_profile_enable(eng, SIMS * prof_steps))
    for i in range(prof_steps):
        step(args.warmup + i)
    n_launch = ctypes.c_int64(0)
    tot_ms = ctypes.c_double(0.0)
    L.check(lib.lz_profile_read(eng, ctypes.byref(n_launch), ctypes.byref(tot_ms)))
    L.check(lib.lz_profile_enable(eng, 0))

    traffic = None
    try:  # HBM bytes per k_chain launch from the PMC passes (FETCH_SIZE x 2 + WRITE_SIZE), see profiles/r01_traffic.json
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
            traffic = json.load(f)["k_chain"]["hbm_bytes_per_launch"]
    except Exception:
        pass
    if rank == 0:
        value = world * ENVS * args.steps / elapsed
        ms_per_step = elapsed / args.steps * 1e3
        avg_us = tot_ms.value / max(n_launch.value, 1) * 1e3
        achieved = (EPS * FLOP_CHAIN) / (avg_us * 1e-6) / 1e12 if n_launch.value else None
        out = {
            "metric": "self-play env-steps/sec @50 sims, 256 envs per GPU (EfficientZero Atari 96x96x4)",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: Atari Pong EfficientZero, obs 4x96x96, 50 sims, "
                                   "256 envs per GPU, A=6, support 601, LSTM 512; synthetic obs, seed-0 random-init weights",
                       "envs_per_gpu": ENVS, "num_simulations": SIMS, "mcts_sims_per_s": value * SIMS,
                       "tiebreak": args.tiebreak, "sub_batches": NS, "whole_step_tflops": value * (SIMS * FLOP_RECURRENT + FLOP_INITIAL) / 1e12,
                       "parallelism": "env-shard x%d" % world},
            "roofline": {"bound": "mfma", "kernel": "k_chain (per root: [tree step of the root: expand + backup + next selection, one wave, prologue] + dynamics conv + 2 residual blocks + 1x1 head convs on the 6x6x64 latent, LDS-resident; 1 launch/simulation; achieved counts only the convolution FLOPs over the whole launch)",
                         "achieved": achieved, "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                         "frac": (achieved / PEAK_FP32_MATRIX_TFLOPS) if achieved else None, "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (rocprofv3 PMC passes of profiles/r01_traffic.json, not re-measured in this run)",
                         "avg_launch_us": avg_us, "launches_timed": n_launch.value,
                         "timing": "HIP event pairs on the engine stream around each launch, %d eager steps run right after the "
                                   "graph-replayed timed region" % prof_steps,
                         "algorithmic_flop_per_launch": EPS * FLOP_CHAIN},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(weights, obs_cpu, [z.tolist() for z in noise_steps[0]])
            out["config"]["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
            try:
                out["deployed_baseline"] = cpu_baseline(weights, obs_cpu, [z.tolist() for z in noise_steps[0]], device="cuda")
                out["config"]["speedup_vs_deployed_baseline"] = value / out["deployed_baseline"]["value"]
            except Exception as e:  # the reported baselines never take the measured line down with them
                out["deployed_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
