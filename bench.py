#!/usr/bin/env python
"""Headline benchmark: self-play env-steps/s (and MCTS sims/s) for EfficientZero Atari, 96x96x4
observations, 50 simulations, 256 parallel envs per MI355X (BASELINE.json configs[1]).

A "step" is one pass of the hot path over one batch of 256 synthetic observations that are already
resident in HBM: initial_inference -> root prepare with Dirichlet noise -> 50 x [select -> recurrent
inference -> expand/backup] on the device -> select_action + the packed env-step rows (action, search
statistics, action mask, to_play, newest observation frame: the GameSegment field set, 36.9 KB per
env-step) written by one kernel -> row headers on the host (the `_forward_collect` contract: what the
collector needs to step its environments).  Weak scaling: every GPU owns its own 256 envs; the only
collective is the all-gather of the rows (lightzero_amd/shard.py), issued asynchronously: it runs
under the next step's representation tower and that step's search is ordered behind it (--gather-fence).

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
PEAK_BF16_MATRIX_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 dense peak (the --fast arm)
PEAK_SPLIT_BF16_TFLOPS = PEAK_BF16_MATRIX_TFLOPS / 6.0   # parity mode since round 5: fp32 operands as three bf16 planes, six plane products per k-step (k_chain_s3)
CFG = dict(num_simulations=SIMS, pb_c_base=19652, pb_c_init=1.25, discount_factor=0.997, value_delta_max=0.01,
           lstm_horizon_len=5, root_noise_weight=0.25, root_dirichlet_alpha=0.3)
MANIFEST = "r06_manifest.json"     # profiles/: rocprofv3 numbers of the roofline kernel + the digest of the sources they were measured on


def _profile_manifest():
    """the committed rocprofv3 summary, if it was measured on the kernel sources this run is built from"""
    try:
        from lightzero_amd.build import csrc_digest
        with open(os.path.join(ROOT, "profiles", MANIFEST)) as f:
            m = json.load(f)
        return m if m.get("csrc_sha256") == csrc_digest() else None
    except Exception:
        return None


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


class ClockSampler(object):
    """The GPU's shader clock while the bench runs (the headline moved 7 % between leases of identical sources; this puts the
    clock the box sustained next to `value`).  Reads the amdgpu hwmon node freq1_input (sclk, Hz) of the card whose PCI address is the HIP
    device's, every 4 ms on a thread; windows are cut out afterwards by wall-clock time.  A box without the node reports why."""

    def __init__(self, device_index=0):
        import glob
        import threading
        self.samples, self.path, self.note = [], None, None
        self._stop = threading.Event()
        self._thr = None
        try:
            import torch
            p = torch.cuda.get_device_properties(device_index)
            bus = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        except Exception as e:
            bus = None
            self.note = "no PCI address for the HIP device: %r" % (e,)
        cands = []
        for f in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")):
            dev = os.path.realpath(f.split("/hwmon/")[0])
            cands.append((f, dev))
        mine = [f for f, dev in cands if bus and bus in dev]
        if mine:
            self.path = mine[0]
        elif cands:
            self.path = [f for f, _ in cands]          # unknown mapping: the busiest card is ours (max over the cards per sample)
            self.note = (self.note or "") + " PCI address %s not found among %d cards: max over all cards" % (bus, len(cands))
        else:
            self.note = "no /sys/class/drm/card*/device/hwmon/*/freq1_input on this box"

    def _read(self):
        def one(f):
            try:
                return int(open(f).read().strip()) / 1e6
            except Exception:
                return 0.0
        return one(self.path) if isinstance(self.path, str) else max(one(f) for f in self.path)

    def start(self):
        import threading
        if not self.path or os.environ.get("LZ_BENCH_NO_CLOCK"):   # (the switch: A/B runs of the sampler's own effect on the timed region)
            return self

        def run():
            while not self._stop.is_set():
                self.samples.append((time.perf_counter(), self._read()))
                self._stop.wait(0.004)
        self._thr = threading.Thread(target=run, daemon=True)
        self._thr.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thr:
            self._thr.join(timeout=1.0)

    def window(self, t_a, t_b):
        v = [m for t, m in self.samples if t_a <= t <= t_b and m > 0]
        if not v:
            return None
        return dict(sclk_mhz_mean=float(np.mean(v)), sclk_mhz_min=float(min(v)), sclk_mhz_max=float(max(v)), samples=len(v))

    def report(self, windows):
        out = {"source": ("amdgpu hwmon freq1_input (sclk) of %s, sampled every 4 ms" % (self.path if isinstance(self.path, str) else "%d cards (max)" % len(self.path)))
               if self.path else None, "note": self.note}
        for name, (t_a, t_b) in windows.items():
            out[name] = self.window(t_a, t_b) if self.path else None
        return out


def cpu_allowance():
    """how many cores this PROCESS may actually use: the scheduler affinity mask and the cgroup CPU quota (cpu.max of cgroup v2 /
    cfs_quota_us of v1) -- a container that sees all 256 hardware threads of the host in /proc/cpuinfo may still be throttled to a
    fraction of them (the signature: the torch-thread sweep collapsing beyond the quota)."""
    out = dict(affinity=len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), quota=None)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            out["quota"] = float(q) / float(p)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                out["quota"] = q / p
        except Exception:
            pass
    return out


def host_cores():
    """(physical cores, hardware threads) of the host: distinct (physical id, core id) pairs of /proc/cpuinfo"""
    threads = os.cpu_count() or 1
    try:
        pairs, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                phys = core = None
        if phys is not None and core is not None:
            pairs.add((phys, core))
        return (len(pairs) or threads), threads
    except Exception:
        return threads, threads


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
    phys, thr = host_cores()
    return dict(value=ENVS / med, unit="env-steps/s", cores=best, host_cores=phys, host_threads=thr, cpu_allowance=cpu_allowance(), kind="port",
                thread_sweep_env_steps_per_s={str(k): round(v, 1) for k, v in sweep.items()},
                batches=len(times), batch_s_min_median_max=[min(times), med, max(times)],
                cores_note="cores = the torch intra-op thread count that won the sweep (the tree and the Python driver are one thread); "
                           "host_cores / host_threads = physical cores / hardware threads of this host",
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


def search_arm(weights, obs_list, steps, warmup, fast=False, seed=7, stamps=True):
    """One secondary arm: the headline's step -- initial inference, device Dirichlet noise + prepare, 50 simulations, select_action + row
    packing, header read-back -- on a model of its own (own engine) carrying ``weights``; observations cycle through ``obs_list``.
    Returns env-steps/s, the search-path depths of the last timed step (lz_roots_get_node_depths: depth of the node each simulation
    expanded == that simulation's search length) and, with ``stamps``, the launch periods from the in-graph s_memrealtime stamps."""
    import torch
    from lightzero_amd import _lib as L, shard
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    lib = L.lib()
    eng = L.new_engine(int(os.environ.get("LOCAL_RANK", "0")))
    model = EfficientZeroModel(action_space_size=ACTIONS, engine=eng, fast_mode=fast).load_state_dict(weights)
    n = obs_list[0].shape[0]
    roots = ez_tree.Roots(n, [list(range(ACTIONS))] * n, action_space_size=ACTIONS, max_simulations=SIMS, engine=eng)
    roots.set_tiebreak(1, seed=seed)
    roots._ensure(ACTIONS)
    to_play = L.i32([-1] * n)
    W, HW = shard.row_width(ACTIONS, FRAME), shard.HEADER + 2 * ACTIONS
    rows = torch.zeros(n, W, device="cuda")
    timestep = np.zeros(n, np.int32)
    torch.cuda.synchronize()

    def one(i):
        L.check(lib.lz_initial_inference(roots._h, obs_list[i % len(obs_list)].data_ptr()))
        L.check(lib.lz_roots_prepare_from_inference_dirichlet(roots._h, CFG["root_noise_weight"], CFG["root_dirichlet_alpha"], to_play))
        L.check(lib.lz_search(roots._h, SIMS, CFG["pb_c_base"], CFG["pb_c_init"], CFG["discount_factor"], CFG["lstm_horizon_len"], CFG["value_delta_max"]))
        timestep[:] = i
        h = np.zeros((n, HW), np.float32); lg = np.zeros((n, ACTIONS), np.float32)
        L.check(lib.lz_roots_collect_rows(roots._h, 1.0, 0, (i * 1315423911 + 5) & (2 ** 62 - 1), None, FRAME, timestep.ctypes.data, rows.data_ptr(), W, h, lg.ctypes.data))
        return h, lg
    for i in range(warmup):
        one(i)
    L.check(lib.lz_engine_synchronize(eng))
    t0 = time.perf_counter()
    for i in range(steps):
        h, lg = one(warmup + i)
    L.check(lib.lz_engine_synchronize(eng))
    dt = time.perf_counter() - t0
    assert (np.array(roots.get_distributions()).sum(1) == SIMS).all()
    out = dict(env_steps_per_s=n * steps / dt, ms_per_step=dt / steps * 1e3, steps=steps, warmup=warmup)
    out.update(_depth_stats(lib, L, roots, n))
    e = np.exp(lg - lg.max(1, keepdims=True))
    out["root_prior_max_prob_mean"] = float((e / e.sum(1, keepdims=True)).max(1).mean())   # before the Dirichlet noise
    if stamps:
        try:
            out.update(_stamp_periods(lib, L, roots, eng, one, warmup, 3))
        except Exception as ex:
            out["stamps_error"] = repr(ex)
    return out


def _depth_stats(lib, L, roots, n):
    """search-path depth (== CSearchResults::search_lens of every simulation) of the search the roots hold"""
    d = np.zeros((n, SIMS), np.int32)
    L.check(lib.lz_roots_get_node_depths(roots._h, SIMS, d.reshape(-1)))
    return dict(search_depth_mean=float(d.mean()), search_depth_max=int(d.max()), search_depth_last10_mean=float(d[:, -10:].mean()),
                search_depth_hist=np.bincount(d.reshape(-1), minlength=1).tolist())


def _stamp_periods(lib, L, roots, eng, one, first_step, nsteps):
    """launch periods of the graph-replayed search from the in-graph s_memrealtime stamps (lz_roots_enable_stamps): the search graph is
    re-captured with the stamp pointers, `nsteps` steps are replayed after two warm-ups"""
    L.check(lib.lz_roots_enable_stamps(roots._h, 1))
    per, lper = [], []
    st = np.zeros((SIMS, 4), np.uint64)
    for i in range(2 + nsteps):
        one(first_step + i)
        L.check(lib.lz_engine_synchronize(eng))
        if i < 2:
            continue
        L.check(lib.lz_roots_read_stamps(roots._h, SIMS, st))
        t = st.astype(np.int64) * 10e-3   # microseconds
        per.append(t[1:SIMS - 1, 2] - t[1:SIMS - 1, 0]); lper.append(t[2:SIMS, 0] - t[1:SIMS - 1, 2])
    L.check(lib.lz_roots_enable_stamps(roots._h, 0))
    per, lper = np.concatenate(per), np.concatenate(lper)
    return dict(chain_period_us=float(per.mean()), lstm_period_us=float(lper.mean()), per_simulation_us=float(per.mean() + lper.mean()))


def fast_mode_arm(weights, obs_list, steps, warmup):
    """The FAST MODE arm beside the parity-mode headline (last arm): the same step on
    EfficientZeroModel(fast_mode=True) (bf16 MFMA products, fp32 accumulation; statistical parity only, tests/test_fast_mode_gpu.py).
    A separate number, never `value`."""
    out = search_arm(weights, obs_list, steps, warmup, fast=True)
    out.update(dtype="bf16 products, f32 accumulation",
               note="FAST MODE arm (EfficientZeroModel(fast_mode=True)): representation tower, recurrent chain and LSTM gate product on bf16 MFMA "
                    "(k_conv_bf, k_chain_b, k_lstm_b); heads, normalisation, cell and tree in fp32; statistical parity only -- reported "
                    "separately from the parity-mode `value` (BASELINE.md section 2, last arm)")
    return out


def depth_sweep(weights, obs_list, steps, warmup, scales=(1, 4, 10)):
    """the section-8d recipe (head layers drawn at N(0, 0.05)) gives near-uniform root priors and the shallowest trees 50
    simulations can build; a trained agent's prior is sharp and its search paths are deeper.  The same step with the policy head's
    and the value head's last layer scaled (lightzero_amd.model.synthetic.sharpen_state_dict): env-steps/s, search-path depth, and the
    chain launch's period (the tree step is that launch's prologue) per scale.  `value` stays on scale 1 = the 8d recipe."""
    from lightzero_amd.model.synthetic import sharpen_state_dict
    arms = {}
    for sc in scales:
        try:
            arms[str(sc)] = search_arm(sharpen_state_dict(weights, float(sc)), obs_list, steps, warmup)
        except Exception as e:   # a secondary arm never takes the measured line down with it
            arms[str(sc)] = {"error": repr(e)}
    base = arms.get("1", {}).get("env_steps_per_s")
    for sc, a in arms.items():
        if base and "env_steps_per_s" in a:
            a["vs_scale_1"] = a["env_steps_per_s"] / base
    return dict(arms=arms, scaled="prediction_network.fc_policy.3 and fc_value.3 (weight and bias) x scale",
                note="scale 1 = SURVEY 8d's recipe (what `value` is measured on); each arm is the headline's step on its own engine, "
                     "%d timed steps after %d warm-ups, stochastic tie-break, device-side Dirichlet noise" % (steps, warmup))


def weight_refresh(model, weights, roots, step, first_step, reps=10):
    """What ONE weight refresh of a collector costs (in the reference the collector searches with the learner's own
    nn.Module, lzero/policy/muzero.py:1049-1061 -- fresh weights are free there): model.load_state_dict of the full EfficientZero Atari
    state_dict on an engine with live roots and a captured search graph -- (a) from host arrays, as a checkpoint arrives, (b) from
    device tensors, as shard.broadcast_state_dict(on_device=True) hands them over.  Wall time of the call plus the engine
    synchronisation behind it, median of `reps`; then one step to show the roots are still live."""
    import torch
    from lightzero_amd import _lib as L
    lib = L.lib()
    eng = model.engine
    from lightzero_amd import shard
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in weights.items()}
    flat = shard.flat_state_dict(weights, "cuda")
    torch.cuda.synchronize()
    model.load_state_dict(flat)   # (the first refresh of a model records its re-layout program: a one-time host pass, not a per-refresh cost)
    out = {}
    for key, sd in (("weight_refresh_ms", weights), ("weight_refresh_device_ms", dev), ("weight_refresh_flat_ms", flat)):
        ts = []
        for _ in range(reps):
            L.check(lib.lz_engine_synchronize(eng))
            t0 = time.perf_counter()
            model.load_state_dict(sd)
            L.check(lib.lz_engine_synchronize(eng))
            ts.append((time.perf_counter() - t0) * 1e3)
        out[key] = float(np.median(ts))
        out[key + "_min_max"] = [float(min(ts)), float(max(ts))]
    step(first_step)   # the same roots, the same captured graph
    L.check(lib.lz_engine_synchronize(eng))
    out["weight_refresh_note"] = ("one model.load_state_dict of the %d-tensor EfficientZero Atari state_dict (%.1f MB) on an engine with live roots, wall time of "
                                  "the call + the engine synchronisation behind it, median of %d: _ms = from host arrays (pinned staging + one upload), _device_ms = "
                                  "from 90 separate device tensors (one concatenation), _flat_ms = from a shard.FlatStateDict (one flat device buffer, what the RCCL "
                                  "broadcast delivers: by pointer); every layout is rebuilt by kernels on the engine's stream (lz_model_refresh_flat), bit-identical "
                                  "to the host re-layout" % (len(weights), sum(v.size for v in weights.values()) * 4 / 1e6, reps))
    return out


def collector_surface(model, groups=1, n_warm_eps=ENVS, n_eps=ENVS + ENVS // 2):
    """env-steps/s through lightzero_amd.worker.MuZeroVectorCollector.collect: the collect loop of muzero_collector.py
    :416-760 -- policy rows, env.step, segment bookkeeping, rollover / pool, device-resident frame stack -- over `groups` synthetic vector
    envs of 256 envs each (frames from a pre-generated pool in pinned host memory, episodes of ~150 steps), while every env is active.
    One group: a step cannot start before the previous step's actions have stepped the environments, so the device waits for the
    host's share of the loop; two groups: the device searches for one while the host steps the other (what a deployment does)."""
    import torch
    from lightzero_amd.policy.efficientzero import EfficientZeroPolicy
    from lightzero_amd.worker import MuZeroVectorCollector
    B, A = ENVS, ACTIONS

    class _Env:
        def __init__(self, seed):
            self.env_num, self.rng, self.k = B, np.random.default_rng(seed), 0
            self.pool = [torch.from_numpy(np.random.default_rng(100 * seed + i).random((B, 1, 96, 96), dtype=np.float32)).pin_memory().numpy() for i in range(4)]
            self.mask, self.tp = np.ones((B, A), np.float32), np.full(B, -1)

        def _obs(self):
            self.k += 1
            return dict(observation=self.pool[self.k % 4], action_mask=self.mask, to_play=self.tp)

        def reset(self):
            return self._obs()

        def step(self, actions, active):
            done = (self.rng.random(B) < 1.0 / 150) & active
            return self._obs(), np.zeros(B, np.float32), done, dict(reset_obs=self._obs(), eval_episode_return=np.zeros(B))
    ccfg = dict(CFG, game_segment_length=400, num_unroll_steps=5, td_steps=5, model=dict(frame_stack_num=4, action_space_size=A))
    envs = [_Env(g) for g in range(groups)]
    pols = [EfficientZeroPolicy(ccfg, model) for _ in range(groups)]
    col = MuZeroVectorCollector(envs[0] if groups == 1 else envs, pols[0] if groups == 1 else pols, ccfg, device="cuda")
    col.collect(n_episode=groups * n_warm_eps)          # warm-up: handles, graphs
    t0 = time.perf_counter()
    l0 = col.total_loop_steps
    col.collect(n_episode=groups * n_eps)
    dt = time.perf_counter() - t0
    return B * (col.total_loop_steps - l0) / dt


def _cpu_worker(spec):
    """one process of the whole-host CPU arm: `idx,nproc,threads,batches` -- this process's contiguous block of the 256 envs, the
    reference pipeline at `threads` torch threads, pinned to its own cores; protocol on stdout / stdin: READY -> go -> one line of JSON"""
    import torch
    idx, nproc, threads, batches = (int(x) for x in spec.split(","))
    try:
        ncpu = os.cpu_count() or 1
        cores = [c for c in range(idx * threads, (idx + 1) * threads) if c < ncpu]
        if cores:
            os.sched_setaffinity(0, cores)
    except Exception:
        pass
    torch.set_num_threads(threads)
    from lightzero_amd.model.synthetic import efficientzero_state_dict
    weights = efficientzero_state_dict(seed=0, action_space_size=ACTIONS)
    g = torch.Generator().manual_seed(1000)
    obs = torch.rand(ENVS, 4, 96, 96, generator=g)
    lo, hi = idx * ENVS // nproc, (idx + 1) * ENVS // nproc
    sub = obs[lo:hi].contiguous()
    noises = [z.tolist() for z in np.random.default_rng(idx).dirichlet([CFG["root_dirichlet_alpha"]] * ACTIONS, size=hi - lo).astype(np.float32)]
    run, kind_tree = _baseline_pipeline(weights, "cpu")
    run(sub[:4], noises)          # warm-up: imports, threads ...
    run(sub, noises)              # ... and this batch size's operator set-up (oneDNN builds its primitives per shape: the first batch of a new
                                  # size is several times slower than the following ones)
    sys.stdout.write("READY\n"); sys.stdout.flush()
    sys.stdin.readline()
    t0 = time.time()
    for _ in range(batches):
        run(sub, noises)
    t1 = time.time()
    sys.stdout.write(json.dumps(dict(idx=idx, envs=hi - lo, t0=t0, t1=t1, kind_tree=kind_tree)) + "\n"); sys.stdout.flush()


def _whole_host_run(nproc, threads, batches, budget_s):
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", "%d,%d,%d,%d" % (i, nproc, threads, batches)],
                              stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, universal_newlines=True,
                              env=dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
             for i in range(nproc)]
    try:
        deadline = time.time() + budget_s
        for p in procs:
            line = p.stdout.readline()
            if line.strip() != "READY" or time.time() > deadline:
                raise RuntimeError("worker did not come up: %r" % line)
        for p in procs:
            p.stdin.write("go\n"); p.stdin.flush()
        res = [json.loads(p.stdout.readline()) for p in procs]
    finally:
        for p in procs:
            try:
                p.stdin.close()
            except Exception:
                pass
        for p in procs:
            try:
                p.wait(timeout=30)
            except Exception:
                p.kill()
    span = max(r["t1"] for r in res) - min(r["t0"] for r in res)
    envs = sum(r["envs"] for r in res)
    return dict(value=envs * batches / span, processes=nproc, threads_per_process=threads, cores=nproc * threads, span_s=span,
                slowest_process_s=max(r["t1"] - r["t0"] for r in res), kind_tree=res[0]["kind_tree"])


def cpu_baseline_whole_host(threads, rank, batches=3, budget_s=150.0):
    """The same reference pipeline with ALL physical cores busy: N processes x T torch threads with N x T = the host's
    physical cores, each process pinned to its own T cores and running the pipeline on its contiguous block of the 256 envs; all start
    together after their warm-ups; rate = 256 envs x batches / (latest end - earliest start).  (N, T) is swept -- T = the thread count
    that won cpu_baseline's single-process sweep, then 4 and 1 (the small per-simulation operators of this model scale better across
    processes than across threads) -- within the time budget; `value` = the best configuration, every configuration is listed."""
    phys, thr = host_cores()
    allow = cpu_allowance()
    budget = int(min(phys, allow["affinity"], allow["quota"] or phys))    # the cores this container may really keep busy
    t_start = time.time()
    runs, seen = [], set()
    for t in (max(threads, 1), 4, 1):
        nproc = max(1, min(budget // t, ENVS // 2))
        if (nproc, t) in seen or (runs and time.time() - t_start > 0.5 * budget_s):
            continue
        seen.add((nproc, t))
        try:
            runs.append(_whole_host_run(nproc, t, batches, budget_s))
        except Exception as e:
            runs.append(dict(processes=nproc, threads_per_process=t, error=repr(e)))
    ok = [r for r in runs if "value" in r]
    if not ok:
        raise RuntimeError("no whole-host configuration completed: %r" % runs)
    best = max(ok, key=lambda r: r["value"])
    return dict(value=best["value"], unit="env-steps/s", cores=best["cores"], processes=best["processes"], threads_per_process=best["threads_per_process"],
                host_cores=phys, host_threads=thr, cpu_allowance=allow, core_budget=budget, kind="port", batches=batches, configurations=runs,
                sample="best of %s (processes x torch threads, each process pinned to its own cores), %d batches of each process's block of the 256 envs x 50 "
                       "sims; %s + restated driver + torch fp32 model; all processes start together; value = envs x batches / (latest end - earliest start); %.0f s"
                       % ([(r["processes"], r["threads_per_process"]) for r in runs], batches, best["kind_tree"], time.time() - t_start))


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
    ap.add_argument("--fast", action="store_true",
                    help="FAST MODE arm (BASELINE.md section 2, last arm): bf16 MFMA for the recurrent chain's convolutions and the LSTM gate "
                         "product, statistical parity only -- a separate number, never the parity-mode headline (the default run)")
    ap.add_argument("--streams", type=int, default=1,
                    help="split the 256 envs of a GPU into this many independent sub-batches, each on its own engine "
                         "(HIP stream): the MFMA-bound conv chain of one overlaps the latency-bound tree / LSTM / head "
                         "kernels of the other")
    ap.add_argument("--tiebreak", choices=["first", "random"], default="random",
                    help="random = the reference's stochastic tie rule (default, like collection); first = parity mode (the other "
                         "arm is timed too and reported in config)")
    ap.add_argument("--noise", choices=["device", "host"], default="device",
                    help="device = Dirichlet root noise drawn by a kernel inside the step (north_star); host = np.random.dirichlet "
                         "inside the step + upload, like efficientzero.py:599-602")
    ap.add_argument("--obs-batches", type=int, default=4, help="distinct synthetic observation batches in HBM, cycled by step")
    ap.add_argument("--no-depth-sweep", action="store_true", help="skip the prior-sharpness arms (config.depth_sweep)")
    ap.add_argument("--cpu-worker", default="", help=argparse.SUPPRESS)   # internal: one process of the whole-host CPU arm
    ap.add_argument("--sustain-s", type=float, default=2.0, help="seconds of extra steps after the timed region for config.sustained_env_steps_per_s")
    ap.add_argument("--sync-gather", action="store_true", help="N > 1: wait for every step's all-gather instead of overlapping it with the next search")
    ap.add_argument("--total-envs", type=int, default=0,
                    help="STRONG scaling (BASELINE configs[3] / [4]: 512 envs over 8 GPUs, 256 over 4): this many envs in total, split into "
                         "contiguous blocks by shard.shard_range (sizes may differ by one); default 0 = weak scaling, 256 envs per GPU")
    ap.add_argument("--refresh-every", type=int, default=0,
                    help="every K steps, INSIDE the timed loop: broadcast the checkpoint's model state_dict from rank 0 (shard.broadcast_state_dict) "
                         "and re-ingest it in place (a collector's weight refresh after a learner update)")
    ap.add_argument("--gather-fence", choices=["tower", "none"], default="tower",
                    help="N > 1, overlapped all-gather: 'tower' (default) lets the collective of step i run under the representation tower of "
                         "step i + 1 and makes the search wait for it -- the search's chain launch is one workgroup per CU on all 256 CUs, so a "
                         "collective kernel holding even one CU during it costs that launch a second round; 'none' = no ordering")
    ap.add_argument("--check-gather", action="store_true", help="after the timed region: every rank verifies the pooled rows block by block")
    args = ap.parse_args()
    if args.cpu_worker:
        _cpu_worker(args.cpu_worker)
        return
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
    # LZ_FORCE_COLLECTIVE=1: run the collective code path with a single rank too (a 1-GPU box then exercises the RCCL branch --
    # process group, asynchronous all-gather, the fence below -- that otherwise only an N > 1 node reaches); never a headline run
    dist_on = world > 1 or bool(os.environ.get("LZ_FORCE_COLLECTIVE"))
    if dist_on and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if dist_on:
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
    if dist_on:  # the weight-refresh path: rank 0's state_dict reaches every rank through one broadcast, then an in-place re-ingest
        weights = shard.broadcast_state_dict(weights, src=0)
    NS = max(1, args.streams)
    if args.total_envs:   # strong scaling: this rank's contiguous block of the global env batch
        blocks = [shard.shard_range(args.total_envs, q, world) for q in range(world)]
        counts = [hi - lo for lo, hi in blocks]
    else:
        counts = [256] * world
    ENVS = counts[rank]   # (shadows the module constant: everything below is per rank)
    NMAX = max(counts)
    assert ENVS > 0 and ENVS % NS == 0
    EPS = ENVS // NS  # envs per sub-batch
    engs, models = [], []
    for k in range(NS):
        e = L.default_engine(device_index) if k == 0 else L.new_engine(device_index)
        engs.append(e)
        models.append(EfficientZeroModel(action_space_size=ACTIONS, engine=e, fast_mode=args.fast).load_state_dict(weights))
    eng = engs[0]
    g = torch.Generator().manual_seed(1000 + rank)
    # NOBS distinct observation batches, resident in HBM, cycled by step: no step searches the batch the previous one searched
    # (one tensor fed to every step); batch 0 is also what the CPU baselines run on
    NOBS = max(1, args.obs_batches)
    obs_pool_cpu = [torch.rand(ENVS, 4, 96, 96, generator=g) for _ in range(NOBS)]
    obs_cpu = obs_pool_cpu[0]
    obs_pool = [o.cuda().contiguous() for o in obs_pool_cpu]
    obs = obs_pool[0]
    rng = np.random.default_rng(rank)
    total = args.warmup + args.steps
    legal = [list(range(ACTIONS))] * EPS
    roots_l = []
    for k in range(NS):
        r = ez_tree.Roots(EPS, legal, action_space_size=ACTIONS, max_simulations=SIMS, engine=engs[k])
        r.set_tiebreak(0 if args.tiebreak == "first" else 1, seed=rank * 16 + k + 1)
        r._ensure(ACTIONS)
        roots_l.append(r)
    to_play = L.i32([-1] * EPS)
    obs_parts = [[o[k * EPS:(k + 1) * EPS].contiguous() for o in obs_pool] for k in range(NS)]
    W = shard.row_width(ACTIONS, FRAME)
    HW = shard.HEADER + 2 * ACTIONS
    # double-buffered: step i's all-gather reads one.  Uneven blocks (strong scaling): every buffer has the largest block's rows, the
    # collective moves equal blocks with no size exchange (all ranks know shard_range) and the consumer strips the padding
    rows_dev = [torch.zeros(NMAX, W, device="cuda") for _ in range(2)]
    gathered = [torch.zeros(world * NMAX, W, device="cuda") for _ in range(2)] if dist_on and backend == "nccl" else None
    gathered_host = [None, None]
    header = np.zeros((ENVS, HW), np.float32)
    logits = np.zeros((ENVS, ACTIONS), np.float32)
    timestep = np.zeros(EPS, np.int32)
    pending = [None, None]
    fence_mode = args.gather_fence
    eng_streams = [torch.cuda.ExternalStream(lib.lz_engine_stream(e)) for e in engs]   # the engines' HIP streams, for stream-ordering only

    host_obs = [None]   # set by the PCIe-inclusive secondary arm below: [sub-batch][obs batch] pinned host tensors

    def step(i):
        buf = i & 1
        if pending[buf] is not None:  # the all-gather that still reads this row buffer (issued two steps ago)
            pending[buf].wait()
            # wait() orders torch's stream behind the collective; the rows are rewritten on the ENGINE's stream, so the host confirms
            # the completion (free: the collective was issued two 3.5 ms steps ago)
            torch.cuda.current_stream().synchronize()
            pending[buf] = None
        for k, r in enumerate(roots_l):  # enqueue everything of every sub-batch before reading anything back
            if host_obs[0] is not None:   # secondary arm only: the observations cross PCIe inside the step (pinned host memory -> HBM)
                L.check(lib.lz_initial_inference_host(r._h, host_obs[0][k][i % NOBS].numpy()))   # (the numpy view of the pinned tensor: same memory)
            else:
                L.check(lib.lz_initial_inference(r._h, obs_parts[k][i % NOBS].data_ptr()))
            if fence_mode == "tower" and pending[buf ^ 1] is not None:
                # the previous step's all-gather has had the tower to itself (thousands of workgroups: a few CUs less cost it a per
                # cent); the search -- 256 workgroups that each need a whole CU -- starts behind it
                with torch.cuda.stream(eng_streams[k]):
                    pending[buf ^ 1].wait()
            if args.noise == "device":   # Dirichlet(0.3) over the legal actions of every root, drawn by a kernel of this step
                L.check(lib.lz_roots_prepare_from_inference_dirichlet(r._h, CFG["root_noise_weight"], CFG["root_dirichlet_alpha"], to_play))
            else:                        # drawn on the host INSIDE the step (one vectorised numpy call) and uploaded
                nz = rng.dirichlet([CFG["root_dirichlet_alpha"]] * ACTIONS, size=EPS).astype(np.float32)
                L.check(lib.lz_roots_prepare_from_inference(r._h, CFG["root_noise_weight"], nz.ctypes.data, to_play))
            L.check(lib.lz_search(r._h, SIMS, CFG["pb_c_base"], CFG["pb_c_init"], CFG["discount_factor"],
                                  CFG["lstm_horizon_len"], CFG["value_delta_max"]))
        for k, r in enumerate(roots_l):  # select_action + packed env-step rows: one kernel, header words back on the host
            timestep[:] = i
            h = np.zeros((EPS, HW), np.float32); lg = np.zeros((EPS, ACTIONS), np.float32)
            L.check(lib.lz_roots_collect_rows(r._h, 1.0, 0, (i * 1315423911 + rank * 97 + k) & (2 ** 62 - 1), None, FRAME, timestep.ctypes.data,
                                              rows_dev[buf][k * EPS:(k + 1) * EPS].data_ptr(), W, h, lg.ctypes.data))
            header[k * EPS:(k + 1) * EPS], logits[k * EPS:(k + 1) * EPS] = h, lg
        if dist_on:  # pool the finished env-step rows of all ranks (RCCL all-gather over xGMI), overlapped with the next search
            if backend == "nccl":
                _, work = shard.all_gather_rows_equal(rows_dev[buf], out=gathered[buf], async_op=not args.sync_gather)
                pending[buf] = work if not args.sync_gather else None
            else:
                gathered_host[buf], _ = shard.all_gather_rows_equal(rows_dev[buf].cpu())
        if args.refresh_every and (i + 1) % args.refresh_every == 0:
            # weight refresh inside the loop: one flat broadcast from rank 0, then the device tensors are overwritten in place
            # (same buffers: the captured search graphs stay valid)
            fresh = shard.broadcast_state_dict(weights_flat, src=0, on_device=True)   # RCCL: one flat device buffer, handed to the library by pointer
            for mdl in models:
                mdl.load_state_dict(fresh)   # device-side re-layout on the engine's stream (lz_model_refresh_flat): no host synchronisation

    # the learner's weights as they reach a collector rank: ONE flat fp32 buffer in HBM (across ranks the in-place RCCL broadcast of that
    # buffer; a learner in the same process flattens its parameters once per update) -- built outside the timed loop, refreshed inside it
    weights_flat = shard.flat_state_dict(weights, "cuda") if args.refresh_every else None
    if weights_flat is not None:   # the first refresh of a model records its re-layout program (one host pass of the packers, ~tens of ms): not a per-refresh cost
        for mdl in models:
            mdl.load_state_dict(weights_flat)

    def drain():
        for b in (0, 1):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    def sync_all():
        drain()
        for e in engs:
            L.check(lib.lz_engine_synchronize(e))
        torch.cuda.synchronize()

    # The interpreter's cyclic collector: a full (generation 2) pass over the ~10^6 objects a process with torch imported holds takes 30-50 ms
    # of HOST time, and it is triggered by allocation counts -- it landed inside the timed region of some runs and not of others (40 timed
    # steps: 60-67 k against 88.6 k env-steps/s on one box; the clock sampler's tuples moved it into the 20-step region: -2.4 %,
    # tools/r06_steps_exp.sh).  Everything allocated so far is moved to the permanent generation (gc.freeze(): the collector stays ON and
    # keeps collecting what the steps allocate); no work of a step is skipped.  LZ_BENCH_NO_GC_FREEZE=1 keeps the old behaviour (A/B).
    import gc
    if not os.environ.get("LZ_BENCH_NO_GC_FREEZE"):
        gc.collect()
        gc.freeze()
    for i in range(args.warmup):
        step(i)
    sync_all()
    if dist_on:
        dist.barrier()
    clocks = ClockSampler(device_index).start() if rank == 0 else None
    clock_windows = {}
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        step(i)
    sync_all()
    my_elapsed = time.perf_counter() - t0
    clock_windows["timed_region"] = (t0, t0 + my_elapsed)
    if dist_on:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [my_elapsed]
    if dist_on:
        dev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([my_elapsed], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [float(x.item()) for x in allr]
    # every step ran all its simulations: child visits of every row sum to 1 with 50 visits behind them
    rows_last = rows_dev[(total - 1) & 1][:ENVS, :HW].cpu().numpy()
    assert np.allclose(rows_last[:, shard.HEADER:shard.HEADER + ACTIONS].sum(1), 1.0, atol=1e-5), "rows carry no search statistics"
    assert np.array_equal(rows_last, header), "host header differs from the device rows"
    dist_chk = np.zeros((EPS, ACTIONS), np.int32); cnt_chk = np.zeros(EPS, np.int32)
    L.check(lib.lz_roots_get_distributions(roots_l[0]._h, dist_chk, cnt_chk))
    assert (dist_chk.sum(1) == SIMS).all(), "search did not run all simulations"
    depth_main = _depth_stats(lib, L, roots_l[0], EPS)   # the trees of the LAST TIMED step (sub-batch 0)
    gather_check = None
    if args.check_gather and dist_on:
        # every rank: the pooled rows of the last step, block q, must be rank q's own rows (exchanged once more, as float64 checksums
        # per row, through a plain all_gather) -- with the padding of uneven blocks stripped
        buf = (total - 1) & 1
        sync_all()
        pooled = (gathered[buf] if backend == "nccl" else gathered_host[buf]).cpu().double()
        mine = rows_dev[buf][:ENVS].cpu().double()
        wts = torch.arange(1, W + 1, dtype=torch.float64)
        sums = torch.zeros(NMAX, dtype=torch.float64)
        sums[:ENVS] = (mine * wts).sum(1)
        cdev = "cuda" if backend == "nccl" else "cpu"
        alls = [torch.zeros(NMAX, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(alls, sums.to(cdev))
        alls = [a.cpu() for a in alls]
        ok = all(torch.equal((pooled[q * NMAX:q * NMAX + counts[q]] * wts).sum(1), alls[q][:counts[q]]) for q in range(world))
        ok = ok and torch.equal(pooled[rank * NMAX:rank * NMAX + ENVS], mine)
        gather_check = "ok" if ok else "MISMATCH"
        assert ok, "the pooled rows differ from the ranks' own rows"
    # ---- secondary arms, rank-local, after the contract's timed region (same process, same roots)
    # (a) sustained: the contract's K = 20 steps are 0.07 s of GPU work, a burst before the clock settles -- keep stepping for
    #     >= --sustain-s seconds and report that rate too
    sustained = None
    if args.sustain_s > 0:
        # the step count comes from the (all-reduced) time of the timed region: identical on every rank, so the row all-gathers match
        n_s = 20 * max(1, int(np.ceil(args.sustain_s / (elapsed / args.steps) / 20)))
        t_s = time.perf_counter()
        for j in range(n_s):
            step(total + j)
        sync_all()
        sustained = dict(env_steps_per_s=sum(counts) * n_s / (time.perf_counter() - t_s), steps=n_s, seconds=time.perf_counter() - t_s)
        clock_windows["sustained"] = (t_s, time.perf_counter())
    # (a2) PCIe-inclusive: the same step with the observation batch handed over in (pinned) HOST memory -- lz_initial_inference_host copies the
    #      37.7 MB over PCIe on the engine's stream in front of the tower.  Never `value` (the contract's inputs are resident in HBM).
    host_obs_rate = None
    if not dist_on and hasattr(lib, "lz_initial_inference_host"):
        host_obs[0] = [[o[k * EPS:(k + 1) * EPS].contiguous().pin_memory() for o in obs_pool_cpu] for k in range(NS)]
        n_w = max(2, NOBS)   # every pinned batch once: the first transfer out of a freshly pinned buffer is slower
        for j in range(n_w):
            step(total + j)
        sync_all()
        t_h = time.perf_counter()
        for j in range(args.steps):
            step(total + n_w + j)
        sync_all()
        host_obs_rate = ENVS * args.steps / (time.perf_counter() - t_h)
        host_obs[0] = None
    if clocks:
        clocks.stop()
    # (b) the other tie-break arm (the parity tests pin "first": deterministic first arg-max; collection uses the reference's
    #     stochastic rule): same K steps after W warm-ups, the search graph re-captured for the other rule
    other = "first" if args.tiebreak == "random" else "random"
    for k, r in enumerate(roots_l):
        r.set_tiebreak(0 if other == "first" else 1, seed=rank * 16 + k + 1)
    for i in range(args.warmup):
        step(i)
    sync_all()
    t_o = time.perf_counter()
    for i in range(args.steps):
        step(i)
    sync_all()
    other_rate = sum(counts) * args.steps / (time.perf_counter() - t_o)   # (rank 0's clock; the ranks run in lock-step through the all-gathers)
    for k, r in enumerate(roots_l):
        r.set_tiebreak(0 if args.tiebreak == "first" else 1, seed=rank * 16 + k + 1)
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

    # THIS run's clock for the roofline kernel, on the benchmarked (graph-replayed) launch sequence: the first workgroup of every
    # chain and LSTM launch stores its s_memrealtime start, the last one its end (100 MHz constant-rate counter; lz_roots_enable_stamps;
    # the search graph is re-captured with the stamp pointers, nothing else changes).  A launch's cost in the stream is the time from
    # its first workgroup's start to the NEXT launch's first start (execution + drain + dispatch of the successor): that period is
    # what `frac` divides by; first workgroup's start -> last workgroup's end of the launch itself is reported beside it.
    stamp = None
    try:
        r0 = roots_l[0]
        L.check(lib.lz_roots_enable_stamps(r0._h, 1))
        for i in range(2):
            step(args.warmup + i)
        per, exe, lper, lexe, sims = [], [], [], [], []
        st = np.zeros((SIMS, 4), np.uint64)
        for i in range(max(prof_steps, 5)):
            step(args.warmup + i)
            drain()
            L.check(lib.lz_engine_synchronize(eng))
            L.check(lib.lz_roots_read_stamps(r0._h, SIMS, st))
            t = st.astype(np.int64) * 10e-3   # microseconds
            # simulations 1 .. S-2: the fused launch (tree step + split-head finish + chain) followed by an LSTM launch and another chain
            per.append(t[1:SIMS - 1, 2] - t[1:SIMS - 1, 0]); exe.append(t[1:SIMS - 1, 1] - t[1:SIMS - 1, 0])
            lper.append(t[2:SIMS, 0] - t[1:SIMS - 1, 2]); lexe.append(t[1:SIMS - 1, 3] - t[1:SIMS - 1, 2])
            sims.append(t[SIMS - 1, 3] - t[0, 0])
        L.check(lib.lz_roots_enable_stamps(r0._h, 0))
        per, exe, lper, lexe = (np.concatenate(x) for x in (per, exe, lper, lexe))
        stamp = dict(chain_period_us=float(per.mean()), chain_exec_us=float(exe.mean()), lstm_period_us=float(lper.mean()), lstm_exec_us=float(lexe.mean()),
                     chain_period_us_min_max=[float(per.min()), float(per.max())], launches=int(per.size),
                     search_us=float(np.mean(sims)), per_simulation_us=float((per.mean() + lper.mean())))
    except Exception as e:   # (an older library without the stamp entry points: the HIP-event pass below still gives a clock)
        stamp = dict(error=repr(e))

    # The committed rocprofv3 summary (profiles/r04_manifest.json) is used only when it was measured on the kernel sources this run is
    # built from (digest of csrc/): then the roofline divides by the profiler's average launch duration and carries the PMC traffic;
    # otherwise by this run's own HIP-event pairs (which include the gaps of eager launches) with traffic null.
    man = _profile_manifest()
    traffic = man["k_chain"]["hbm_bytes_per_launch"] if man else None
    if rank == 0:
        value = sum(counts) * args.steps / elapsed
        ms_per_step = elapsed / args.steps * 1e3
        ev_us = tot_ms.value / max(n_launch.value, 1) * 1e3
        have_stamp = bool(stamp) and "chain_period_us" in stamp and EPS == ENVS
        # frac comes from THIS run's clock (in-graph stamps); the committed profile's figure rides along as frac_profile
        avg_us = stamp["chain_period_us"] if have_stamp else ev_us
        clock = ("this run: s_memrealtime stamps inside the graph-replayed search (first workgroup start of the launch -> first workgroup "
                 "start of the next launch; %d launches)" % stamp["launches"]) if have_stamp \
            else "this run: HIP event pairs around eagerly launched steps (gaps included)"
        achieved = (EPS * FLOP_CHAIN) / (avg_us * 1e-6) / 1e12 if avg_us else None
        prof_us = man["k_chain"]["rocprof_avg_us"] if (man and EPS == 256) else None
        achieved_prof = (EPS * FLOP_CHAIN) / (prof_us * 1e-6) / 1e12 if prof_us else None
        knobs = sorted(k for k in os.environ if k.startswith("LZ_"))
        out = {
            "metric": "self-play env-steps/sec @50 sims, 256 envs per GPU (EfficientZero Atari 96x96x4; fp32 parity mode: network products as six exact bf16-plane products per k-step)",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if args.total_envs else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: Atari Pong EfficientZero, obs 4x96x96, 50 sims, "
                                   "256 envs per GPU, A=6, support 601, LSTM 512; synthetic obs, seed-0 random-init weights; dtype f32 = binary32 operands and "
                                   "accumulators, the 3x3 convolutions' products evaluated as six exact bf16 x bf16 plane products per k-step (config.arithmetic)",
                       "envs_per_gpu": ENVS, "envs_per_rank": counts, "total_envs": sum(counts), "num_simulations": SIMS,
                       "weight_refresh_every": args.refresh_every, "gather_check": gather_check, "mcts_sims_per_s": value * SIMS,
                       "tiebreak": args.tiebreak, "tiebreak_%s_env_steps_per_s" % other: other_rate,
                       "root_noise": "drawn on the device inside every step (lz_roots_prepare_from_inference_dirichlet)" if args.noise == "device"
                                     else "np.random dirichlet drawn inside every step, uploaded",
                       "sustained_env_steps_per_s": sustained["env_steps_per_s"] if sustained else None,
                       "host_obs_env_steps_per_s": host_obs_rate,
                       "host_obs_note": "the same K steps with the observation batch handed over in pinned HOST memory (lz_initial_inference_host: %.1f MB over PCIe per step, "
                                        "on the engine's stream in front of the tower); reported beside `value`, never as it" % (ENVS * 4 * 96 * 96 * 4 / 1e6),
                       "sustained_steps": sustained["steps"] if sustained else 0, "sustained_seconds": sustained["seconds"] if sustained else 0.0,
                       "sub_batches": NS, "whole_step_tflops": value * (SIMS * FLOP_RECURRENT + FLOP_INITIAL) / 1e12,
                       "parallelism": "env-shard x%d" % world, "rccl_ranks": world if backend == "nccl" else 0, "collective_backend": backend if dist_on else None,
                       "per_rank_env_steps_per_s": [counts[q] * args.steps / t for q, t in enumerate(per_rank)],
                       "row_bytes_per_env_step": W * 4, "all_gather_bytes_per_step_per_rank": (world - 1) * NMAX * W * 4 if world > 1 else 0,
                       "all_gather_overlapped": bool(dist_on and backend == "nccl" and not args.sync_gather),
                       "all_gather_fence": fence_mode if (dist_on and backend == "nccl" and not args.sync_gather) else None,
                       "arithmetic": "tree: binary32 (bit-exact with the reference's ctree); network: binary32 operands as three exact bf16 planes, six "
                                     "plane products per k-step accumulated in binary32 on the bf16 matrix pipe (tower, recurrent chain), binary32 "
                                     "matrix instructions elsewhere (LSTM, heads) -- within 1e-5 (1 + |x|) of the reference modules",
                       "gpu_clock": clocks.report(clock_windows) if clocks else None,
                       "debug_knobs": knobs},
            "roofline": {"bound": "mfma", "kernel": "k_chain_s3 (per root: [tree step of the root: expand + backup + next selection, one wave; the previous leaf's head MLPs on the other seven: prologue] + dynamics conv + 2 residual blocks + 1x1 head convs on the 6x6x64 latent, LDS-resident; 3x3 convolutions in the direct form as SPLIT-bf16 products on v_mfma_f32_16x16x32_bf16: every fp32 operand is the exact sum of three bf16 planes, six of the nine plane products are accumulated in fp32 -- fp32-level accuracy at 6 bf16 matrix FLOPs per algorithmic FLOP; 1 launch/simulation; achieved = the ALGORITHMIC convolution FLOPs of SURVEY 8d over the whole launch; peak = the bf16 dense peak / 6, the rate at which this form can deliver fp32-accurate FLOPs)",
                         "achieved": achieved, "peak": PEAK_SPLIT_BF16_TFLOPS, "unit": "TFLOP/s",
                         "peak_note": "2500 TFLOP/s bf16 dense / 6 plane products = %.1f TFLOP/s of fp32-accurate work; against the fp32 matrix pipe's own peak (157.3 TFLOP/s, what k_chain_w -- LZ_CHAIN_NO_SPLIT=1 -- is priced on) the same achieved rate is frac_vs_fp32_matrix_peak" % PEAK_SPLIT_BF16_TFLOPS,
                         "executed_bf16_tflops": (achieved * 6.0 * 48.0 / 36.0) if achieved else None,
                         "executed_note": "matrix work actually issued: 6 products x 48 / 36 (the 36 pixels of the latent fill three 16-pixel tiles) -- of the 2500 TFLOP/s bf16 peak",
                         "frac_vs_fp32_matrix_peak": (achieved / PEAK_FP32_MATRIX_TFLOPS) if achieved else None,
                         "frac": (achieved / PEAK_SPLIT_BF16_TFLOPS) if achieved else None, "traffic": traffic,
                         "traffic_unit": ("HBM bytes per launch: rocprofv3 PMC passes (FETCH_SIZE x 2 + WRITE_SIZE) of profiles/%s" % MANIFEST) if man
                                         else "null: no committed PMC pass of these kernel sources",
                         "avg_launch_us": avg_us, "clock": clock,
                         "avg_exec_us": stamp.get("chain_exec_us") if stamp else None,
                         "exec_note": "avg_exec_us = first workgroup start -> last workgroup end of the launch (no dispatch gap); frac uses avg_launch_us",
                         "achieved_exec": ((EPS * FLOP_CHAIN) / (stamp["chain_exec_us"] * 1e-6) / 1e12) if have_stamp else None,
                         "lstm_launch_us": stamp.get("lstm_period_us") if stamp else None, "lstm_exec_us": stamp.get("lstm_exec_us") if stamp else None,
                         "per_simulation_us": stamp.get("per_simulation_us") if stamp else None, "stamps": stamp,
                         "achieved_profiled": achieved_prof, "frac_profile": (achieved_prof / PEAK_SPLIT_BF16_TFLOPS) if achieved_prof else None,
                         "avg_launch_us_profile": prof_us,
                         "profile": ("rocprofv3 --kernel-trace average of profiles/%s (measured on these kernel sources: csrc digest matches)" % MANIFEST) if prof_us
                                    else "no committed profile of these kernel sources",
                         "avg_launch_us_hip_events": ev_us, "launches_timed": n_launch.value,
                         "timing": "in-graph s_memrealtime stamps over %d graph-replayed steps right after the timed region (this run); HIP event "
                                   "pairs around each launch of %d eagerly launched steps as a cross-check" % (max(prof_steps, 5), prof_steps),
                         "algorithmic_flop_per_launch": EPS * FLOP_CHAIN},
        }
        if args.fast:
            out["metric"] += " -- FAST MODE"
            out["dtype"] = "bf16"
            out["config"]["mode"] = ("fast mode (EfficientZeroModel(fast_mode=True), lz_model_cfg.precision = 1): bf16 MFMA with fp32 accumulation for the recurrent "
                                     "chain's 3x3 convolutions (k_chain_b) and the LSTM gate product (k_lstm_b); representation tower, normalisation, cell, heads "
                                     "and tree in fp32; statistical parity only (tests/test_fast_mode_gpu.py) -- NOT the parity-mode headline")
            rf = out["roofline"]
            rf["kernel"] = "k_chain_b (the same launch as k_chain_s3 with ONE bf16 product per k-step: the 3x3 convolutions in the direct form on v_mfma_f32_16x16x32_bf16)"
            rf["peak"] = PEAK_BF16_MATRIX_TFLOPS
            rf["frac"] = (achieved / PEAK_BF16_MATRIX_TFLOPS) if achieved else None
            rf["bound"] = "latency (tree step + weight stream of 74 KB per layer and CU); the matrix pipe is idle most of the launch"
            for k in ("traffic", "achieved_profiled", "frac_profile", "avg_launch_us_profile"):
                rf[k] = None
            for k in ("peak_note", "executed_bf16_tflops", "executed_note", "frac_vs_fp32_matrix_peak"):
                rf.pop(k, None)
            rf["traffic_unit"] = rf["profile"] = "fast mode: no committed PMC / rocprofv3 pass"
        out["config"].update(search_depth_mean=depth_main["search_depth_mean"], search_depth_max=depth_main["search_depth_max"],
                             search_depth_last10_mean=depth_main["search_depth_last10_mean"], search_depth_hist=depth_main["search_depth_hist"],
                             search_depth_note="depth of the node each of the 50 simulations expanded (== its search-path length), all roots of "
                                               "the last timed step; observations: %d distinct batches cycled by step" % NOBS,
                             obs_batches=NOBS)
        if world == 1 and not args.fast and NS == 1:
            try:
                out["fast_mode"] = fast_mode_arm(weights, obs_pool, args.steps, args.warmup)
            except Exception as e:   # a secondary arm never takes the measured line down with it
                out["fast_mode"] = {"error": repr(e)}
            if not args.no_depth_sweep:
                out["config"]["depth_sweep"] = depth_sweep(weights, obs_pool, args.steps, args.warmup)
            try:
                out["config"].update(weight_refresh(models[0], weights, roots_l[0], step, args.warmup))
            except Exception as e:
                out["config"]["weight_refresh_ms"] = repr(e)
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["config"]["policy_surface_env_steps_per_s"] = policy_surface(models[0], obs)
            except Exception as e:
                out["config"]["policy_surface_env_steps_per_s"] = repr(e)
            for key, ng in (("collector_env_steps_per_s", 1), ("collector_2groups_env_steps_per_s", 2)):
                try:
                    out["config"][key] = collector_surface(models[0], ng)
                except Exception as e:
                    out["config"][key] = repr(e)
            noises0 = [z.tolist() for z in rng.dirichlet([CFG["root_dirichlet_alpha"]] * ACTIONS, size=ENVS).astype(np.float32)]
            out["cpu_baseline"] = cpu_baseline(weights, obs_cpu, noises0)
            out["config"]["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
            try:
                out["cpu_baseline_whole_host"] = cpu_baseline_whole_host(out["cpu_baseline"]["cores"], rank)
                out["config"]["speedup_vs_cpu_baseline_whole_host"] = value / out["cpu_baseline_whole_host"]["value"]
            except Exception as e:
                out["cpu_baseline_whole_host"] = {"error": repr(e)}
            try:
                out["deployed_baseline"] = deployed_baseline(weights, obs_cpu, noises0)
                out["config"]["speedup_vs_deployed_baseline_min_median_max"] = [value / x for x in out["deployed_baseline"]["env_steps_per_s_min_median_max"][::-1]]
            except Exception as e:  # the reported baselines never take the measured line down with them
                out["deployed_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
