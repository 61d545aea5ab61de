#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out/r05n; mkdir -p $R
for k in 0 2 4 7; do
cd /tmp && export TMPDIR=/tmp
LZ_CONV_SKEW=$k timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$R/stats$k -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --sustain-s 0 --no-depth-sweep > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find $R -name "*kernel_trace.csv" -delete
echo "== skew $k"
python - <<P
import csv,glob
f=glob.glob('gpurun_out/r05n/stats$k/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_conv_s3' in r['Name']: print('%-50s avg %8.1f us' % (r['Name'].replace('void (anonymous namespace)::','')[:50], float(r['AverageNs'])/1e3))
P
done
