"""Copies the judged summaries of a profiling run into profiles/ (tracked):

    gpurun -- '<the rocprofv3 commands in DESIGN.md section 4, outputs under gpurun_out/<run>/{stats,pmc_sq,pmc_fetch,pmc_write},
               plus python bench.py > gpurun_out/<run>/bench_n1.json>'
    python tools/refresh_profiles.py gpurun_out/<run> r01

writes profiles/<tag>_final_kernel_stats.csv, <tag>_bench_n1.json, <tag>_traffic.json, <tag>_pmc_summary.txt, <tag>_parity.json (the
measured worst-case floating-point differences of the run's GPU tests, per kernel variant) and <tag>_manifest.json: the digest of the
kernel sources everything above was measured on (lightzero_amd.build.csrc_digest, written ON THE GPU BOX by tools/profile_run.sh),
the git commit this tool ran at, and the roofline kernel's rocprofv3 numbers -- what bench.py reads when the digest matches.
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    """'void (anonymous namespace)::k_x<1, 2>(args...)' -> 'k_x<1, 2>'"""
    name = name.replace("void (anonymous namespace)::", "")
    if ">(" in name:
        return name[:name.index(">(") + 1]
    return name.split("(")[0]


def main():
    run, tag = sys.argv[1], sys.argv[2]
    prof = os.path.join(ROOT, "profiles")
    stats = glob.glob(os.path.join(run, "stats", "**", "*kernel_stats.csv"), recursive=True)[0]
    shutil.copy(stats, os.path.join(prof, "%s_final_kernel_stats.csv" % tag))
    line = open(os.path.join(run, "bench_n1.json")).read().strip().splitlines()[-1]
    bench = json.loads(line)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in ("pmc_sq", "pmc_fetch", "pmc_write"):
        for f in glob.glob(os.path.join(run, d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(stats))}

    def find(key, table):
        for name in table:
            if key in name:
                return table[name]
        return None

    def mean(key, counter):
        t = find(key, acc)
        v = t.get(counter) if t else None
        return round(sum(v) / len(v), 2) if v else None

    # the roofline kernel: the parity-mode chain instantiation with the largest total time (the fused tree-step + chain launch):
    # k_chain_s3 since round 5 (k_chain_w when the run had LZ_CHAIN_NO_SPLIT=1)
    tot = {r["Name"]: float(r["TotalDurationNs"]) for r in csv.DictReader(open(stats))}
    fused = max((n for n in tot if "k_chain_s3" in n or "k_chain_w" in n), key=lambda n: tot[n])
    fused = short(fused)
    f, w = mean(fused, "FETCH_SIZE"), mean(fused, "WRITE_SIZE")
    out = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and rocprofv3 --kernel-trace --pmc WRITE_SIZE (two separate passes) -- "
                   "python bench.py --steps 1 --warmup 1 --no-cpu-baseline, MI355X; per-dispatch means in KB as reported "
                   "(tools/refresh_profiles.py). gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B "
                   "request of a wide coalesced read, so read bytes = 2 x FETCH_SIZE; WRITE_SIZE taken as reported (uncalibrated).",
           "k_chain": {"fetch_size_kb": f, "write_size_kb": w, "hbm_bytes_per_launch": int(round((2 * f + w) * 1024)),
                       "kernel": fused,
                       "note": "the root's prologue (tree step on wave 0; split heads: the previous simulation's head MLPs finished on waves 1-7 "
                               "from the LSTM launch's 3.1 MB of first-layer partial sums + 0.16 MB of second-layer weights per XCD) + the "
                               "split-bf16 convolution chain. reads: latent gather 2.36 MB + 1.11 MB of weight planes (5 layers x 64 x 64 x 9 x "
                               "3 bf16) once per XCD L2 (8 x: L2 does not survive a kernel boundary) + the staged trees; writes: next latent "
                               "2.36 MB + head-conv outputs 1.77 MB + tree write-through. Algorithmic bytes of the chain 7.8 MB (activations "
                               "in and out + the fp32 weights once)."}}
    # every other kernel of the step, under the name the profiler printed (looking kernels up by last round's template
    # signature left the LSTM's and the row kernel's entries null): one entry per kernel family = its instantiation with the largest total time
    fams = {}
    for n in tot:
        sn = short(n)
        fam = sn.split("<")[0]
        if not fam.startswith("k_") or sn == fused:
            continue
        if fam not in fams or tot[n] > tot[fams[fam][0]]:
            fams[fam] = (n, sn)
    for fam, (n, sn) in sorted(fams.items()):
        fs, ws = mean(sn, "FETCH_SIZE"), mean(sn, "WRITE_SIZE")
        if fs is None and ws is None:
            continue
        out[sn] = {"fetch_size_kb": fs, "write_size_kb": ws, "avg_us": round(dur.get(n, 0.0), 2),
                   "hbm_bytes_per_launch": int(round((2 * (fs or 0.0) + (ws or 0.0)) * 1024))}
    json.dump(out, open(os.path.join(prof, "%s_traffic.json" % tag), "w"), indent=1)
    # the bench line of the same run read the PREVIOUS traffic file (bench.py takes roofline.traffic from profiles/): carry this run's
    # counters instead, so that the committed line and the committed counters belong together
    d_chain = find(fused, dur)
    rl = bench["roofline"]
    rl["traffic"] = out["k_chain"]["hbm_bytes_per_launch"]
    rl["traffic_unit"] = ("HBM bytes per launch (rocprofv3 PMC passes of the same profiling run, profiles/%s_traffic.json; "
                          "patched in by tools/refresh_profiles.py)" % tag)
    # the line carries its own clock (in-graph stamps of that run: roofline.avg_launch_us / frac); the profiler's average duration of the
    # same profiling run rides along as the *_profile fields (what bench.py fills in by itself once the manifest below is committed)
    rl["avg_launch_us_profile"] = d_chain
    rl["achieved_profiled"] = rl["algorithmic_flop_per_launch"] / (d_chain * 1e-6) / 1e12
    rl["frac_profile"] = rl["achieved_profiled"] / rl["peak"]
    rl["profile"] = "rocprofv3 --kernel-trace average of the same profiling run (profiles/%s_final_kernel_stats.csv; patched in by tools/refresh_profiles.py)" % tag
    open(os.path.join(prof, "%s_bench_n1.json" % tag), "w").write(json.dumps(bench) + "\n")
    # MFMA utilisation of the roofline kernel: busy cycles per SIMD over the launch duration at the sustained clock
    busy = mean(fused, "SQ_VALU_MFMA_BUSY_CYCLES")
    d_us = d_chain
    hdr = ["# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY",
           "#   + separate passes --pmc FETCH_SIZE / --pmc WRITE_SIZE   -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline   (MI355X)",
           "# per-dispatch means.  Counters are summed over the 8 XCDs / 1024 SIMDs; SQ_WAVE_CYCLES / SQ_ACTIVE_INST_ANY count quad-cycles;",
           "# SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD (32 per v_mfma_f32_16x16x4_f32, 16 per v_mfma_f32_16x16x32_bf16);",
           "# k_chain_s3 runs 324 16x16x32 bf16 MFMAs per SIMD and layer (9 taps x 2 K halves x 6 plane products x 3 pixel tiles): 5,184 busy cycles;",
           "# the fp32 Winograd form it replaced (k_chain_w) needed 6,144, the direct fp32 form 10,368 for the same convolution.",
           "# FETCH_SIZE / WRITE_SIZE in KB (FETCH_SIZE x 2 = bytes read on gfx950, see %s_traffic.json)." % tag]
    if busy and d_us:
        per_simd = busy / 1024.0
        hdr.append("# %s: %.2f M MFMA-busy cycles / 1024 SIMDs = %.1f k cycles per SIMD; launch %.1f us (%s_final_kernel_stats.csv) x ~2.0 GHz"
                   % (fused, busy / 1e6, per_simd / 1e3, d_us, tag))
        hdr.append("#   = %.0f k cycles -> MFMA pipe busy %.0f %% of the launch (which includes the tree-step prologue)."
                   % (d_us * 2.0, 100.0 * per_simd / (d_us * 2000.0)))
    body = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_pmc.py")] +
                          [os.path.join(run, d) for d in ("pmc_sq", "pmc_fetch", "pmc_write")],
                          stdout=subprocess.PIPE, universal_newlines=True, check=True).stdout
    open(os.path.join(prof, "%s_pmc_summary.txt" % tag), "w").write("\n".join(hdr) + "\n" + body)
    # ---- measured parity of the run's GPU tests, all kernel variants in one file
    par = {}
    for f in sorted(glob.glob(os.path.join(run, "parity", "*.json"))):
        d = json.load(open(f))
        par[d["variant"]] = d["tests"]
        bounds, unit = d["bounds"], d["unit"]
    if par:
        worst = {}
        own = {}   # tests that assert bounds of their own (the randomised model sweep): worst per class and the loosest bound used
        for v, tests in par.items():
            for t, classes in tests.items():
                tgt = own if (isinstance(classes.get("bounds"), dict) or t.startswith("fuzz/")) else worst
                for k, x in classes.items():
                    if k in bounds and isinstance(x, float):
                        tgt[k] = max(tgt.get(k, 0.0), x)
                if tgt is own:
                    for k, x in (classes.get("bounds") or {}).items():
                        own["bound_" + k] = max(own.get("bound_" + k, 0.0), x)
        json.dump({"_how": "worst |device - reference| / (1 + |reference|) per tensor class, recorded by the GPU tests of the profiling run "
                           "(tests/parity_record.py) for the default kernels and for every alternative path behind an LZ_* switch; "
                           "bounds = what the tests assert (north_star's 1e-5 before the inverse scalar transform; 3e-4 after it, DESIGN.md section 6)",
                   "unit": unit, "bounds": bounds, "worst_over_everything": worst,
                   "worst_over_tests_with_their_own_bounds": own, "variants": par},
                  open(os.path.join(prof, "%s_parity.json" % tag), "w"), indent=1, sort_keys=True)
    # ---- manifest
    man = json.load(open(os.path.join(run, "manifest.json")))
    try:
        man["git_head_at_refresh"] = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, stdout=subprocess.PIPE, universal_newlines=True).stdout.strip()
        man["csrc_dirty_at_refresh"] = bool(subprocess.run(["git", "status", "--porcelain", "lightzero_amd/csrc", "include"], cwd=ROOT, stdout=subprocess.PIPE,
                                                             universal_newlines=True).stdout.strip())
    except Exception:
        pass
    sys.path.insert(0, ROOT)
    from lightzero_amd.build import csrc_digest
    man["csrc_sha256_here"] = csrc_digest()
    man["k_chain"] = {"kernel": fused, "rocprof_avg_us": d_chain, "hbm_bytes_per_launch": out["k_chain"]["hbm_bytes_per_launch"],
                      "mfma_busy_cycles_per_simd": (busy / 1024.0) if busy else None}
    man["kernel_avg_us"] = {short(k): round(v, 2) for k, v in dur.items() if v > 3.0 and "k_" in k}
    # the other BASELINE configurations, measured by tools/config_lines.py into <run>/cfg/
    for f in sorted(glob.glob(os.path.join(run, "cfg", "cfg*.json"))):
        shutil.copy(f, os.path.join(prof, "%s_%s" % (tag, os.path.basename(f))))
    if os.path.isdir(os.path.join(run, "tree", "stats")):   # tools/tree_traffic.sh ran in the same profiling run
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tree_traffic.py"), os.path.join(run, "tree"), tag], check=True, stdout=subprocess.DEVNULL)
    man["files"] = sorted(f for f in os.listdir(prof) if f.startswith(tag + "_"))
    json.dump(man, open(os.path.join(prof, "%s_manifest.json" % tag), "w"), indent=1, sort_keys=True)
    if man["csrc_sha256"] != man["csrc_sha256_here"]:
        print("WARNING: the kernel sources here differ from the ones profiled on the GPU box")
    r = bench["roofline"]
    print("value %.0f env-steps/s, %.3f ms/step; roofline %.1f TFLOP/s frac %.3f (%.1f us/launch by the run's stamps); cpu_baseline %.1f; chain %.2f us by rocprofv3, traffic %d B"
          % (bench["value"], bench["ms_per_step"], r["achieved"], r["frac"], r["avg_launch_us"], bench["cpu_baseline"]["value"], d_us or 0,
             out["k_chain"]["hbm_bytes_per_launch"]))


if __name__ == "__main__":
    main()
