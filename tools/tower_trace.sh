#!/bin/bash
# per-dispatch durations of the representation tower's kernels (one env-step), from a rocprofv3 kernel trace
R=$GRAFT_REPO_ROOT/gpurun_out/tower_trace; mkdir -p $R
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain-s 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os
f = glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/tower_trace/**/*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last occurrence of k_conv_first = start of the last step's tower
# (the default bench line also runs the fast-mode arm, whose first layer writes bf16: "...true>"; LZ_TOOL_FAST=1 picks that one)
fast = bool(os.environ.get("LZ_TOOL_FAST"))
i0 = max(i for i, n in enumerate(names) if "k_conv_first" in n and (("true>" in n) == fast))
for r in rows[i0:i0 + 22]:
    n = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
    print("%-50s %8.1f us  grid %s wg %s" % (n[:50], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
PY
rm -rf $R
