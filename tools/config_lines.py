#!/usr/bin/env python
"""Measured lines for the BASELINE.json configurations that are NOT the headline (configs[0], [2], [3], [4]) on one MI355X, in the
schema of bench.py's line: every configuration is timed twice -- plainly (value / ms_per_step) and under
`rocprofv3 --kernel-trace --stats` (per-kernel table; the roofline object is built from the kernel with the largest share of the
step: algorithmic FLOP per launch / its average duration against the fp32-matrix peak for the MFMA kernels, "latency" for the tree
kernels, whose HBM traffic is a per-cent of the peak).  Writes <out>/cfg{0,2,3,4}.json; tools/refresh_profiles.py copies them to
profiles/rNN_cfgK.json.

    python tools/config_lines.py gpurun_out/<run> [name ...]         (on the GPU box; names = a subset of the configurations)
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = 157.3   # TFLOP/s, dense fp32 matrix (MI355X_MICROARCH.md)
PEAK_SPLIT = 2500.0 / 6.0   # k_chain_s3: fp32 operands as three bf16 planes, six plane products per k-step on the bf16 pipe (2500 TFLOP/s dense)


def chain_peak(kernel):
    """(peak, extra fields) for a chain kernel's algorithmic fp32 FLOPs: the split-bf16 form is priced on the bf16 peak / 6"""
    if "k_chain_s3" in kernel:
        return PEAK_SPLIT, dict(peak_note="bf16 dense peak / 6 plane products (split-bf16 parity-mode chain)", fp32_matrix_peak=PEAK)
    return PEAK, {}

# name -> (BASELINE.json index, tool command, steps under the profiler, per-root algorithmic FLOP of the chain launch | None)
CHAIN_ATARI_MZ = 2 * 36 * 64 * (68 + 4 * 64) * 9 + 2 * 36 * 64 * 48          # dyn conv 68 -> 64 (A = 4) + 4 convs 64 -> 64 + three 1x1 64 -> 16
CHAIN_GO = 2 * 81 * 64 * (146 + 4 * 64) * 9 + 2 * 81 * 64 * 48               # dyn conv (64 + 82) -> 64 on 9x9 + 4 convs + three 1x1
CHAIN_ATARI64_EZ = 2 * 64 * 64 * (70 + 4 * 64) * 9 + 2 * 64 * 64 * 48        # EfficientZero on the 8x8 latent (A = 6): dyn conv 70 -> 64 + 4 convs + three 1x1
CONFIGS = {
    "cfg0": dict(index=0, cmd=["tools/bench_mlp_configs.py", "--config", "0", "--steps", "50"], prof_steps="20",
                 workload="BASELINE.json configs[0]: CartPole-v0 MuZero, MuZeroModelMLP (obs 4, latent 128), 25 sims, 8 envs"),
    "cfg2": dict(index=2, cmd=["tools/bench_conv_configs.py", "--envs", "1024", "--sims", "400", "--steps", "4", "--warmup", "1"], prof_steps="1",
                 chain_flop=CHAIN_ATARI_MZ, workload="BASELINE.json configs[2]: Atari Breakout MuZero, obs 4x96x96, 400 sims, 1024 envs, A = 4"),
    # the same batch as two 512-root sub-batches on two engines / HIP streams: one half's latency-bound HBM tree step
    # runs under the other half's chain as far as the register file allows (the chain launch holds 248 VGPRs x 2 waves per SIMD)
    "cfg2_2streams": dict(index=2, cmd=["tools/bench_conv_configs.py", "--envs", "1024", "--sims", "400", "--steps", "4", "--warmup", "1", "--streams", "2"], prof_steps="1",
                          chain_flop=CHAIN_ATARI_MZ, workload="BASELINE.json configs[2] as two 512-root sub-batches on two streams: Atari Breakout MuZero, 400 sims, 1024 envs, A = 4"),
    "cfg3": dict(index=3, cmd=["tools/bench_conv_configs.py", "--go", "--envs", "64", "--sims", "200", "--steps", "10", "--warmup", "2"], prof_steps="3",
                 chain_flop=CHAIN_GO, workload="BASELINE.json configs[3], one GPU's share: Go 9x9 MuZero, obs 17x9x9, A = 82, 200 sims, 64 of the 512 envs (8 GPUs)"),
    "cfg3_256": dict(index=3, cmd=["tools/bench_conv_configs.py", "--go", "--envs", "256", "--sims", "200", "--steps", "6", "--warmup", "1"], prof_steps="2",
                     chain_flop=CHAIN_GO, workload="BASELINE.json configs[3] on 2 GPUs instead of 8: Go 9x9 MuZero, 256 of the 512 envs on this GPU (one chain workgroup per CU)"),
    "cfg3_all": dict(index=3, cmd=["tools/bench_conv_configs.py", "--go", "--envs", "512", "--sims", "200", "--steps", "4", "--warmup", "1"], prof_steps="1",
                     chain_flop=CHAIN_GO, workload="BASELINE.json configs[3], all 512 envs on one GPU: Go 9x9 MuZero, obs 17x9x9, A = 82, 200 sims"),
    # FAST MODE arms (bf16 matrix products; statistical parity only) and the reference's shipped Atari shape (64x64 frames -> 8x8 latent)
    "cfg2_fast": dict(index=2, cmd=["tools/bench_conv_configs.py", "--envs", "1024", "--sims", "400", "--steps", "4", "--warmup", "1", "--fast"], prof_steps="1", dtype="bf16",
                      workload="FAST MODE arm of BASELINE.json configs[2]: Atari Breakout MuZero, obs 4x96x96, 400 sims, 1024 envs, A = 4"),
    "cfg_atari64": dict(index=1, cmd=["tools/bench_conv_configs.py", "--family", "ez", "--obs", "64", "--envs", "256", "--sims", "50", "--actions", "6", "--steps", "20"], prof_steps="5", chain_flop=CHAIN_ATARI64_EZ,
                        workload="the reference's shipped Atari EfficientZero shape (zoo/atari/config/atari_efficientzero_config.py: 4x64x64 frames -> 8x8 latent, supports (-50, 51)), 50 sims, 256 envs, A = 6"),
    "cfg_atari64_fast": dict(index=1, cmd=["tools/bench_conv_configs.py", "--family", "ez", "--obs", "64", "--envs", "256", "--sims", "50", "--actions", "6", "--steps", "20", "--fast"], prof_steps="5", dtype="bf16",
                             workload="FAST MODE arm of the shipped Atari EfficientZero shape (4x64x64 frames -> 8x8 latent), 50 sims, 256 envs, A = 6"),
    "cfg4": dict(index=4, cmd=["tools/bench_mlp_configs.py", "--config", "4", "--envs", "64", "--steps", "50"], prof_steps="20",
                 workload="BASELINE.json configs[4], one GPU's share: DMC cartpole-swingup Sampled EfficientZero (obs 5, action dim 1, K = 20), 50 sims, 64 of the 256 envs (4 GPUs)"),
    "cfg4_all": dict(index=4, cmd=["tools/bench_mlp_configs.py", "--config", "4", "--envs", "256", "--steps", "50"], prof_steps="20",
                     workload="BASELINE.json configs[4], all 256 envs on one GPU: Sampled EfficientZero (obs 5, action dim 1, K = 20), 50 sims"),
}


def run_tool(cmd):
    r = subprocess.run([sys.executable] + cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError("%s failed: %s" % (cmd, r.stderr[-1500:]))
    return json.loads(lines[-1])


def kernel_table(cmd, outdir, steps):
    cmd = list(cmd)
    if "--steps" in cmd:
        cmd[cmd.index("--steps") + 1] = steps
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", outdir, "--", sys.executable] + [os.path.join(ROOT, cmd[0])] + cmd[1:],
                   cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    f = glob.glob(os.path.join(outdir, "**", "*kernel_stats.csv"), recursive=True)
    for t in glob.glob(os.path.join(outdir, "**", "*kernel_trace.csv"), recursive=True):
        os.remove(t)
    if not f:
        return []
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
    return [dict(kernel=r["Name"].replace("void (anonymous namespace)::", "").split("(")[0], calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3,
                 share=float(r["TotalDurationNs"]) / tot) for r in rows[:8]]


def main():
    out = os.path.abspath(sys.argv[1])
    os.makedirs(out, exist_ok=True)
    from lightzero_amd.build import csrc_digest
    only = sys.argv[2:]
    for name, c in CONFIGS.items():
        if only and name not in only:
            continue
        try:
            line = run_tool(c["cmd"])
            table = kernel_table(c["cmd"], os.path.join(out, "prof_" + name), c["prof_steps"])
        except Exception as e:   # one configuration never takes the others down
            json.dump({"error": repr(e)}, open(os.path.join(out, name + ".json"), "w"))
            continue
        top = table[0] if table else None
        roof = None
        if top:
            roof = dict(kernel=top["kernel"], avg_launch_us=top["avg_us"], share_of_gpu_time=top["share"], traffic=None)
            if "k_chain" in top["kernel"] and c.get("chain_flop"):
                flop = c["chain_flop"] * line["envs"] // max(1, line.get("sub_batches", 1))
                ach = flop / (top["avg_us"] * 1e-6) / 1e12
                pk, extra = chain_peak(top["kernel"])
                roof.update(bound="mfma", algorithmic_flop_per_launch=flop, achieved=ach, peak=pk, unit="TFLOP/s", frac=ach / pk,
                            clock="rocprofv3 --kernel-trace average of this run", **extra)
                if extra:
                    roof["frac_vs_fp32_matrix_peak"] = ach / PEAK
            elif "bench_mlp_configs" in c["cmd"][0]:
                # a stated bound for the launch-bound families.  The floor of a simulation on ONE stream is its launch count x the
                # period of an empty launch in a captured graph (1.7 us: tools/ubench/handoff.py measured 1.6-1.8 us per phase of 256 one-per-CU
                # workgroups as separate launches) -- every launch of these families is a dependent step (dense level -> dense level -> LSTM ->
                # row finisher -> tree step), their weights (<= 0.26 MB per layer, 3.6 MB per simulation) stay in L2 / MALL and their matrix work is
                # ~2 MFLOP per layer.  frac = floor / measured time per simulation: how far the launch sequence is from being purely boundary-bound.
                sims_total = float(c["prof_steps"]) * line["num_simulations"]
                per_sim = sum(k["calls"] for k in table if not k["kernel"].startswith("__amd")) / sims_total
                launches = int(round(per_sim))
                floor_us = launches * 1.7
                meas_us = line["ms_per_step"] * 1e3 / line["num_simulations"]
                roof.update(bound="launch", unit="us per simulation", launches_per_simulation=launches, launch_period_us=1.7,
                            peak=floor_us, achieved=meas_us, frac=floor_us / meas_us if meas_us else None,
                            note="launch-bound family: floor = launches per simulation x 1.7 us (the period of an empty dependent launch on one stream, "
                                 "tools/ubench/handoff.py); achieved = measured ms_per_step / simulations (includes the initial inference's share); "
                                 "frac = floor / achieved.  A dense layer is <= 0.26 MB of weights and ~2 MFLOP, a tree step one wavefront of "
                                 "dependent instructions per root: neither the HBM nor the MFMA roofline is within two orders of magnitude")
            else:
                roof.update(bound="latency", achieved=None, peak=None, frac=None, note="latency-bound kernel at the top of this configuration's table")
        d = {"metric": "self-play env-steps/sec (search only: initial inference -> prepare -> fused search -> read-back, inputs in HBM)",
             "value": line["env_steps_per_s"], "unit": "env-steps/s", "n_gpus": 1, "ms_per_step": line["ms_per_step"], "higher_is_better": True,
             "dtype": c.get("dtype", "f32"), "data": "synthetic", "vs_baseline": None,
             "config": {"workload": c["workload"], "baseline_config_index": c["index"], "envs": line["envs"], "num_simulations": line["num_simulations"],
                        "mcts_sims_per_s": line["mcts_sims_per_s"], "tool": " ".join(c["cmd"])},
             "roofline": roof, "kernels": table, "csrc_sha256": csrc_digest()}
        json.dump(d, open(os.path.join(out, name + ".json"), "w"), indent=1)
        print(name, "%.2f ms/step, %.0f env-steps/s; top kernel %s %.1f us (%.0f %%)%s" % (
            line["ms_per_step"], line["env_steps_per_s"], top["kernel"][:40] if top else "-", top["avg_us"] if top else 0, 100 * (top["share"] if top else 0),
            (" frac %.3f" % roof["frac"]) if roof and roof.get("frac") else ""))


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    main()
