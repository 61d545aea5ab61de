#!/bin/bash
# round 5, GPU call 6: split-bf16 tower (k_conv_s3): numerics first, then the A/B against the Winograd tower
R=gpurun_out/r05f
mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_nn_golden_gpu.py tests/test_nn_gpu.py tests/test_weight_refresh_gpu.py tests/test_obs64_gpu.py tests/test_nn_fuzz_gpu.py tests/test_e2e_cfg1_gpu.py -m gpu -q -p no:cacheprovider > $R/pytest.log 2>&1
tail -15 $R/pytest.log
for i in 1 2; do
for v in "LZ_CONV_NO_SPLIT=1" "LZ_NOTHING=0"; do
  env $v timeout 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --sustain-s 0 --no-depth-sweep 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-20s value %.0f ms %.3f  per-sim %.2f  search_us %.0f' % ('$v', d['value'], d['ms_per_step'], r['per_simulation_us'], r['stamps']['search_us']))"
done; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$R/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 15 --warmup 2 --no-cpu-baseline --sustain-s 0 --no-depth-sweep > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find $R -name "*kernel_trace.csv" -delete
python - <<'P'
import csv,glob
f=glob.glob('gpurun_out/r05f/stats/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print('%-70s calls %5s avg %8.1f us  %5s %%' % (r['Name'].replace('void (anonymous namespace)::','')[:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
P
