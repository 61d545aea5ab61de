#!/bin/bash
# LDS bank-conflict counters per kernel (rocprofv3 --pmc, its own pass): SQ_LDS_BANK_CONFLICT = extra LDS cycles, SQ_LDS_IDX_ACTIVE = all LDS-array cycles
R=$GRAFT_REPO_ROOT/gpurun_out/pmc_lds; mkdir -p $R
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $R -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --sustain-s 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os, collections
f = glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/pmc_lds/**/*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_LDS": cnt[k] += 1
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0))[:8]:
    a, c, i = v.get("SQ_LDS_IDX_ACTIVE", 0), v.get("SQ_LDS_BANK_CONFLICT", 0), v.get("SQ_INSTS_LDS", 0)
    n = max(cnt[k], 1)
    print("%-46s launches %4d  LDS active %12.0f  conflict %12.0f (%.1f %%)  LDS insts %10.0f  per launch: active %.0f conflict %.0f" % (k[:46], n, a, c, 100 * c / max(a, 1), i, a / n, c / n))
PY
find $R -name "*kernel_trace.csv" -delete
