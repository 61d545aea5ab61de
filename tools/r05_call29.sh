#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out/r05p; mkdir -p $R
timeout 900 python -m pytest tests/test_nn_golden_gpu.py tests/test_nn_gpu.py tests/test_obs64_gpu.py tests/test_nn_fuzz_gpu.py -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$R/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --sustain-s 0 --no-depth-sweep > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find $R -name "*kernel_trace.csv" -delete
python - <<P
import csv,glob
f=glob.glob('gpurun_out/r05p/stats/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_conv' in r['Name'] or 'k_avgpool' in r['Name']: print('%-60s calls %4s avg %8.1f us' % (r['Name'].replace('void (anonymous namespace)::','')[:60], r['Calls'], float(r['AverageNs'])/1e3))
P
for v in "LZ_NOTHING=0"; do echo "== bench $v"; env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain-s 0 --no-depth-sweep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['per_simulation_us'])"; done
