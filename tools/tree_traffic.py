"""profiles/<tag>_tree_traffic.json from a tools/tree_traffic.sh run: per tree kernel of BASELINE configs[2] the launch count, average
duration (rocprofv3 --kernel-trace --stats) and HBM bytes per launch (separate --pmc FETCH_SIZE / WRITE_SIZE passes; gfx950: read bytes =
2 x FETCH_SIZE, MI355X_MICROARCH.md) -> GB/s and the fraction of the ~8 TB/s peak; beside them the algorithmic bytes of the step.

    python tools/tree_traffic.py gpurun_out/<run> r04
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK_GBS = 8000.0


def short(name):
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return name[:name.index(">(") + 1] if ">(" in name else name.split("(")[0]


def main():
    run, tag = sys.argv[1], sys.argv[2]
    stats = glob.glob(os.path.join(run, "stats", "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = {short(r["Name"]): r for r in csv.DictReader(open(stats))}
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in ("pmc_fetch", "pmc_write"):
        for f in glob.glob(os.path.join(run, d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    B, S, A = 1024, 400, 4
    out = {"_how": "BASELINE configs[2] (Atari MuZero conv, 1024 roots x 400 simulations, A = 4), python tools/bench_conv_configs.py --envs 1024 "
                   "--sims 400 --steps 1 --warmup 1 under rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate "
                   "passes (tools/tree_traffic.sh); per-dispatch means, KB as reported; read bytes = 2 x FETCH_SIZE on gfx950 "
                   "(MI355X_MICROARCH.md, HBM section), WRITE_SIZE as reported.  hbm_gb_per_s = bytes per launch / average duration.",
           "reference_functions": "cbatch_backpropagate / cbackpropagate lzero/mcts/ctree/ctree_muzero/lib/cnode.cpp:419-478, cbatch_traverse :754-825 "
                                  "(EfficientZero twins: ctree_efficientzero/lib/cnode.cpp:482-601, 886-963)",
           "manifest": json.load(open(os.path.join(run, "manifest.json"))) if os.path.exists(os.path.join(run, "manifest.json")) else None,
           "kernels": {}}
    try:
        line = [l for l in open(os.path.join(run, "line.json")).read().splitlines() if l.startswith("{")]
        out["bench_line_under_the_profiler"] = json.loads(line[-1]) if line else None
    except Exception:
        pass
    tot = sum(float(r["TotalDurationNs"]) for r in rows.values()) or 1.0
    for name, r in sorted(rows.items(), key=lambda kv: -float(kv[1]["TotalDurationNs"])):
        if not any(k in name for k in ("k_backprop", "k_traverse", "k_tree_step_wg", "k_prepare", "k_chain", "k_heads", "k_readout", "k_collect")):
            continue
        f = acc.get(name, {}).get("FETCH_SIZE"); w = acc.get(name, {}).get("WRITE_SIZE")
        fk = sum(f) / len(f) if f else None; wk = sum(w) / len(w) if w else None
        us = float(r["AverageNs"]) / 1e3
        ent = dict(launches=int(r["Calls"]), avg_us=round(us, 2), share_of_gpu_time=round(float(r["TotalDurationNs"]) / tot, 4),
                   fetch_size_kb=round(fk, 1) if fk is not None else None, write_size_kb=round(wk, 1) if wk is not None else None)
        if fk is not None and wk is not None:
            b = (2 * fk + wk) * 1024
            ent["hbm_bytes_per_launch"] = int(b)
            ent["hbm_gb_per_s"] = round(b / (us * 1e-6) / 1e9, 1)
            ent["frac_of_hbm_peak"] = round(b / (us * 1e-6) / 1e9 / PEAK_GBS, 4)
        if "k_tree_step_wg" in name:
            ent["algorithmic_bytes_note"] = ("round 5: a workgroup per root scores EVERY expanded node each simulation (thread = node): per root and simulation "
                                             "nodes x (A x 20 B of edge + child records + 24 B of node record) read once + the path's statistics written; at "
                                             "simulation s that is (s + 1) x 104 B for A = 4 -> ~21 KB per root at the mean tree size of a 400-simulation search, "
                                             "~21 MB per launch for %d roots; DESIGN.md section 3.1c" % B)
        if "k_backprop_traverse" in name:
            ent["algorithmic_bytes_note"] = ("per root and simulation: depth x (A x 20 B of edge + child records read + 16 B of statistics written) with "
                                             "depth ~13..54 over these simulations -> ~3 MB per launch for %d roots; measured traffic above that is "
                                             "128-byte lines around 16-byte records; the step is ONE wavefront of dependent instructions per root "
                                             "(latency-bound), DESIGN.md section 3.1" % B)
        out["kernels"][name] = ent
    p = os.path.join(ROOT, "profiles", "%s_tree_traffic.json" % tag)
    json.dump(out, open(p, "w"), indent=1)
    print(p)
    for k, v in out["kernels"].items():
        print("%-50s %s" % (k[:50], {a: b for a, b in v.items() if a != "algorithmic_bytes_note"}))


if __name__ == "__main__":
    main()
