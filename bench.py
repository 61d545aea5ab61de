#!/usr/bin/env python
"""Headline benchmark: self-play env-steps/s (and MCTS sims/s) for EfficientZero Atari, 96x96x4
observations, 50 simulations, 256 parallel envs per MI355X (BASELINE.json configs[1]).

A "step" is one pass of the hot path over one batch of 256 synthetic observations that are already
resident in HBM: initial_inference -> root prepare with Dirichlet noise -> 50 x [select -> recurrent
inference -> expand/backup] on the device -> select_action + the packed env-step rows (action, search
statistics, action mask, to_play, newest observation frame: the GameSegment field set, 36.9 KB per
env-step) written by one kernel -> row headers on the host (the `_forward_collect` contract: what the
collector needs to step its environments).  Weak scaling: every GPU owns its own 256 envs; the only
collective is the all-gather of the rows (lightzero_amd/shard.py), issued asynchronously so that it
overlaps the next step's search.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8                       # re-executes itself under torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS, SIMS, ACTIONS = 256, 50, 6
FRAME = 96 * 96                  # image_channel = 1: the newest frame of the 4-frame stack
FLOP_RECURRENT = 18381312        # per env per simulation
FLOP_INITIAL = 292222912         # per env-step
FLOP_CHAIN = 2 * 36 * 64 * (70 + 4 * 64) * 9 + 2 * 36 * 64 * 48  # dyn conv 70->64 + 4 convs 64->64 + three 1x1 64->16 = 13,741,056 per env
PEAK_FP32_MATRIX_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32 dense peak
CFG = dict(num_simulations=SIMS, pb_c_base=19652, pb_c_init=1.25, discount_factor=0.997, value_delta_max=0.01,
           lstm_horizon_len=5, root_noise_weight=0.25, root_dirichlet_alpha=0.3)
TRAFFIC_FILE = "r02_traffic.json"


def _reference_model(weights):
    """the oracle's torch restatement of EfficientZeroModel carrying the benchmark's weights (baseline legs only)"""
    import torch
    from oracle import torch_models as tm
    m = tm.EfficientZeroModel(action_space_size=ACTIONS)
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()}, strict=False)
    assert not missing.unexpected_keys and all(k.endswith("num_batches_tracked") for k in missing.missing_keys), missing
    return m.eval()


def _baseline_pipeline(weights, device):
    from oracle import build_ref, ctree as octree, search as osearch
    ref_model = _reference_model(weights)
    if device != "cpu":
        ref_model = ref_model.to(device)
    mods = build_ref.load("stock")
    kind_tree = "reference ctree (oracle/_ref/stock)" if mods else "C restatement of the ctree (oracle/ctree_oracle.c)"
    tree = mods[0] if mods else octree.ez_tree
    kw = {} if mods else dict(roots_kwargs=dict(action_space_size=ACTIONS, max_simulations=SIMS))
    kw["device"] = device

    def run(obs, noises):
        n = obs.shape[0]
        t0 = time.perf_counter()
        osearch.ez_forward_collect(tree, ref_model, obs, [list(range(ACTIONS))] * n, noises[:n], [-1] * n, CFG, **kw)
        return time.perf_counter() - t0
    return run, kind_tree


def cpu_baseline(weights, obs_cpu, noises, budget_s=30.0):
    """The reference pipeline on the host cores: reference ctree (compiled from its own sources when oracle/_ref is present,
    else the C restatement) + restated Python driver + torch fp32 model.  METHOD: the torch thread count is swept over
    {8, 16, 32, 64, all hardware threads up to 128} on one 64-env sub-batch each after a 4-env warm-up (a setting whose warm-up is already
    3x slower than the best one so far is recorded as such and skipped: with one thread per hardware thread the tiny 6x6
    convolutions of this model run ~1000x slower than at 16 threads); the fastest setting then runs full 256-env x 50-sim
    env-step batches until 5 are timed or the budget is spent (never fewer than 2); value = envs / median batch time."""
    import torch
    run, kind_tree = _baseline_pipeline(weights, "cpu")
    allc = os.cpu_count() or torch.get_num_threads()
    # capped at 128: one torch thread per hardware thread of the 256-thread GPU box was measured once at 0.1 env-steps/s
    # (gpurun_out/r2d/bench.json, round 2) -- ten minutes for a single 64-env batch
    cands = sorted({c for c in (8, 16, 32, 64, min(allc, 128)) if c <= allc})
    sub = obs_cpu[:64]
    sweep, best_warm = {}, None
    for c in cands:
        torch.set_num_threads(c)
        warm = run(sub[:4], noises)   # also the warm-up at this thread count
        if best_warm is not None and warm > 3.0 * best_warm:
            sweep[c] = 4 / warm  # hopeless: not worth a 64-env batch
            continue
        best_warm = warm if best_warm is None else min(best_warm, warm)
        sweep[c] = 64 / run(sub, noises)
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    times, used = [], 0.0
    while len(times) < 5 and (used < budget_s or len(times) < 2):
        times.append(run(obs_cpu, noises))
        used += times[-1]
    torch.set_num_threads(allc)
    med = float(np.median(times))
    return dict(value=ENVS / med, unit="env-steps/s", cores=best, kind="port",
                thread_sweep_env_steps_per_s={str(k): round(v, 1) for k, v in sweep.items()},
                batches=len(times), batch_s_min_median_max=[min(times), med, max(times)],
                sample="%d full env-step batches (256 envs x 50 sims) at the best of a torch-thread sweep %s (64-env sub-batch each); "
                       "%s + restated EfficientZeroMCTSCtree.search driver + torch fp32 model; median batch time; %.1f s" %
                       (len(times), cands, kind_tree, used))


def deployed_baseline(weights, obs_cpu, noises, batches=10):
    """SURVEY.md 8d's second arm, the reference as deployed with cuda=True: same driver and host ctree, torch model on the MI355X
    through stock PyTorch-ROCm (MIOpen / rocBLAS).  10 timed batches after two warm-ups (MIOpen solver search), min / median / max."""
    run, kind_tree = _baseline_pipeline(weights, "cuda")
    obs = obs_cpu.to("cuda")
    run(obs[:32], noises)
    run(obs, noises)
    times = [run(obs, noises) for _ in range(batches)]
    med = float(np.median(times))
    return dict(value=ENVS / med, unit="env-steps/s", cores=1, kind="port", batches=batches,
                env_steps_per_s_min_median_max=[ENVS / max(times), ENVS / med, ENVS / min(times)],
                sample="%d full env-step batches after warm-up; %s on 1 host thread + restated driver + torch fp32 model on the "
                       "MI355X via stock PyTorch-ROCm (the reference with cuda=True); median batch time; %.1f s" % (batches, kind_tree, sum(times)))


def policy_surface(model, obs, steps=10):
    """env-steps/s through EfficientZeroPolicy._forward_collect (device select_action): search + the Python glue of the policy surface"""
    from lightzero_amd.policy.efficientzero import EfficientZeroPolicy
    pol = EfficientZeroPolicy(dict(CFG, device_select_action=True), model)
    mask = np.ones((ENVS, ACTIONS), np.float32)
    for _ in range(3):
        pol._forward_collect(obs, action_mask=mask, temperature=1.0, to_play=[-1] * ENVS)
    t0 = time.perf_counter()
    for _ in range(steps):
        pol._forward_collect(obs, action_mask=mask, temperature=1.0, to_play=[-1] * ENVS)
    return ENVS * steps / (time.perf_counter() - t0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


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
    ap.add_argument("--sync-gather", action="store_true", help="N > 1: wait for every step's all-gather instead of overlapping it with the next search")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (one rank per GPU, rendezvous on 127.0.0.1)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    ndev = torch.cuda.device_count()
    # one rank per GPU over RCCL; with fewer devices than ranks (plumbing runs on a 1-GPU box) the ranks share devices and the
    # collective goes through gloo on host copies of the rows -- reported as such, never a headline number
    backend = "nccl" if ndev >= world else "gloo"
    device_index = local_rank % max(ndev, 1)
    torch.cuda.set_device(device_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group("gloo")

    from lightzero_amd import _lib as L, shard
    from lightzero_amd.model.synthetic import efficientzero_state_dict  # seeded random-init weights (no checkpoints offline)
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    lib = L.lib()
    weights = efficientzero_state_dict(seed=0, action_space_size=ACTIONS)
    if world > 1:  # the weight-refresh path: rank 0's state_dict reaches every rank through one broadcast, then an in-place re-ingest
        weights = shard.broadcast_state_dict(weights, src=0)
    NS = max(1, args.streams)
    assert ENVS % NS == 0
    EPS = ENVS // NS  # envs per sub-batch
    engs, models = [], []
    for k in range(NS):
        e = L.default_engine(device_index) if k == 0 else L.new_engine(device_index)
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
    W = shard.row_width(ACTIONS, FRAME)
    HW = shard.HEADER + 2 * ACTIONS
    rows_dev = [torch.zeros(ENVS, W, device="cuda") for _ in range(2)]          # double-buffered: step i's all-gather reads one
    gathered = [torch.zeros(world * ENVS, W, device="cuda") for _ in range(2)] if world > 1 and backend == "nccl" else None
    header = np.zeros((ENVS, HW), np.float32)
    logits = np.zeros((ENVS, ACTIONS), np.float32)
    timestep = np.zeros(EPS, np.int32)
    pending = [None, None]

    def step(i):
        buf = i & 1
        if pending[buf] is not None:  # the all-gather that still reads this row buffer (issued two steps ago)
            pending[buf].wait()
            # wait() orders torch's stream behind the collective; the rows are rewritten on the ENGINE's stream, so the host confirms
            # the completion (free: the collective was issued two 3.5 ms steps ago)
            torch.cuda.current_stream().synchronize()
            pending[buf] = None
        for k, r in enumerate(roots_l):  # enqueue everything of every sub-batch before reading anything back
            L.check(lib.lz_initial_inference(r._h, obs_parts[k].data_ptr()))
            nz = np.ascontiguousarray(noise_steps[i][k * EPS:(k + 1) * EPS])
            L.check(lib.lz_roots_prepare_from_inference(r._h, CFG["root_noise_weight"], nz.ctypes.data, to_play))
            L.check(lib.lz_search(r._h, SIMS, CFG["pb_c_base"], CFG["pb_c_init"], CFG["discount_factor"],
                                  CFG["lstm_horizon_len"], CFG["value_delta_max"]))
        for k, r in enumerate(roots_l):  # select_action + packed env-step rows: one kernel, header words back on the host
            timestep[:] = i
            h = np.zeros((EPS, HW), np.float32); lg = np.zeros((EPS, ACTIONS), np.float32)
            L.check(lib.lz_roots_collect_rows(r._h, 1.0, 0, (i * 1315423911 + rank * 97 + k) & (2 ** 62 - 1), None, FRAME, timestep.ctypes.data,
                                              rows_dev[buf][k * EPS:(k + 1) * EPS].data_ptr(), W, h, lg.ctypes.data))
            header[k * EPS:(k + 1) * EPS], logits[k * EPS:(k + 1) * EPS] = h, lg
        if world > 1:  # pool the finished env-step rows of all ranks (RCCL all-gather over xGMI), overlapped with the next search
            if backend == "nccl":
                _, work = shard.all_gather_rows_equal(rows_dev[buf], out=gathered[buf], async_op=not args.sync_gather)
                pending[buf] = work if not args.sync_gather else None
            else:
                shard.all_gather_rows(rows_dev[buf].cpu())

    def drain():
        for b in (0, 1):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    for i in range(args.warmup):
        step(i)
    drain()
    torch.cuda.synchronize()
    for e in engs:
        L.check(lib.lz_engine_synchronize(e))
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        step(i)
    drain()
    for e in engs:
        L.check(lib.lz_engine_synchronize(e))
    torch.cuda.synchronize()
    my_elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [my_elapsed]
    if world > 1:
        dev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([my_elapsed], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [float(x.item()) for x in allr]
    # every step ran all its simulations: child visits of every row sum to 1 with 50 visits behind them
    rows_last = rows_dev[(total - 1) & 1][:, :HW].cpu().numpy()
    assert np.allclose(rows_last[:, shard.HEADER:shard.HEADER + ACTIONS].sum(1), 1.0, atol=1e-5), "rows carry no search statistics"
    assert np.array_equal(rows_last, header), "host header differs from the device rows"
    dist_chk = np.zeros((EPS, ACTIONS), np.int32); cnt_chk = np.zeros(EPS, np.int32)
    L.check(lib.lz_roots_get_distributions(roots_l[0]._h, dist_chk, cnt_chk))
    assert (dist_chk.sum(1) == SIMS).all(), "search did not run all simulations"
    # Roofline pass: the timed region replays the search from a captured HIP graph, which cannot carry event
    # records, so the dominant kernel is timed right after it, same process and inputs, with HIP event pairs recorded
    # on the engine stream around every k_chain launch of `prof_steps` eagerly launched steps.
    prof_steps = min(args.steps, 5)
    L.check(lib.lz_profile_enable(eng, SIMS * prof_steps))
    for i in range(prof_steps):
        step(args.warmup + i)
    drain()
    n_launch = ctypes.c_int64(0)
    tot_ms = ctypes.c_double(0.0)
    L.check(lib.lz_profile_read(eng, ctypes.byref(n_launch), ctypes.byref(tot_ms)))
    L.check(lib.lz_profile_enable(eng, 0))

    traffic = None
    try:  # HBM bytes per k_chain launch from the PMC passes (FETCH_SIZE / WRITE_SIZE), see profiles/
        with open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)) as f:
            traffic = json.load(f)["k_chain"]["hbm_bytes_per_launch"]
    except Exception:
        pass
    if rank == 0:
        value = world * ENVS * args.steps / elapsed
        ms_per_step = elapsed / args.steps * 1e3
        avg_us = tot_ms.value / max(n_launch.value, 1) * 1e3
        achieved = (EPS * FLOP_CHAIN) / (avg_us * 1e-6) / 1e12 if n_launch.value else None
        knobs = sorted(k for k in os.environ if k.startswith("LZ_"))
        out = {
            "metric": "self-play env-steps/sec @50 sims, 256 envs per GPU (EfficientZero Atari 96x96x4)",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: Atari Pong EfficientZero, obs 4x96x96, 50 sims, "
                                   "256 envs per GPU, A=6, support 601, LSTM 512; synthetic obs, seed-0 random-init weights",
                       "envs_per_gpu": ENVS, "num_simulations": SIMS, "mcts_sims_per_s": value * SIMS,
                       "tiebreak": args.tiebreak, "sub_batches": NS, "whole_step_tflops": value * (SIMS * FLOP_RECURRENT + FLOP_INITIAL) / 1e12,
                       "parallelism": "env-shard x%d" % world, "rccl_ranks": world if backend == "nccl" else 0, "collective_backend": backend if world > 1 else None,
                       "per_rank_env_steps_per_s": [ENVS * args.steps / t for t in per_rank],
                       "row_bytes_per_env_step": W * 4, "all_gather_bytes_per_step_per_rank": (world - 1) * ENVS * W * 4 if world > 1 else 0,
                       "all_gather_overlapped": bool(world > 1 and backend == "nccl" and not args.sync_gather),
                       "debug_knobs": knobs},
            "roofline": {"bound": "mfma", "kernel": "k_chain_w (per root: [tree step of the root: expand + backup + next selection, one wave, prologue] + dynamics conv + 2 residual blocks + 1x1 head convs on the 6x6x64 latent, LDS-resident, 3x3 convolutions by Winograd F(2x2,3x3) on v_mfma_f32_4x4x1; 1 launch/simulation; achieved = the ALGORITHMIC (direct-form) convolution FLOPs of SURVEY 8d over the whole launch -- the kernel executes 0.59x as many matrix cycles for them)",
                         "achieved": achieved, "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                         "frac": (achieved / PEAK_FP32_MATRIX_TFLOPS) if achieved else None, "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (rocprofv3 PMC passes of profiles/%s, not re-measured in this run)" % TRAFFIC_FILE,
                         "avg_launch_us": avg_us, "launches_timed": n_launch.value,
                         "timing": "HIP event pairs on the engine stream around each launch, %d eager steps run right after the "
                                   "graph-replayed timed region" % prof_steps,
                         "algorithmic_flop_per_launch": EPS * FLOP_CHAIN},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["config"]["policy_surface_env_steps_per_s"] = policy_surface(models[0], obs)
            except Exception as e:
                out["config"]["policy_surface_env_steps_per_s"] = repr(e)
            noises0 = [z.tolist() for z in noise_steps[0]]
            out["cpu_baseline"] = cpu_baseline(weights, obs_cpu, noises0)
            out["config"]["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
            try:
                out["deployed_baseline"] = deployed_baseline(weights, obs_cpu, noises0)
                out["config"]["speedup_vs_deployed_baseline_min_median_max"] = [value / x for x in out["deployed_baseline"]["env_steps_per_s_min_median_max"][::-1]]
            except Exception as e:  # the reported baselines never take the measured line down with them
                out["deployed_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
