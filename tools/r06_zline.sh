#!/bin/bash
# round 6: the zero pixel of out-of-image taps read at the lane's own bank quad (k_chain_s3; liblz_mi355_base.so = the build before) -- suites, same-box A/B,
# LDS bank-conflict counter of both builds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/zline
timeout 900 python -m pytest tests/test_nn_golden_gpu.py tests/test_nn_gpu.py tests/test_exact_replay_gpu.py tests/test_e2e_cfg1_gpu.py tests/test_determinism_gpu.py tests/test_muzero_gpu.py tests/test_kernel_variants_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4
bash tools/ab_lib2.sh lightzero_amd/liblz_mi355_base.so 2>&1 | tee gpurun_out/zline/ab.txt
cd /tmp && export TMPDIR=/tmp
for lib in liblz_mi355.so liblz_mi355_base.so; do
  LZ_MI355_LIB=$GRAFT_REPO_ROOT/lightzero_amd/$lib timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/zline/pmc_$lib -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-depth-sweep --sustain-s 0 > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/zline/pmc_$lib/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_chain_s3<6, 6, 1, true>" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$lib", {k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/zline/pmc.txt
find $GRAFT_REPO_ROOT/gpurun_out/zline -name "*.csv" -delete
