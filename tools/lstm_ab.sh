cd /tmp && export TMPDIR=/tmp
export LZ_MI355_LIB=$GRAFT_REPO_ROOT/lightzero_amd/liblz_mi355_dbg.so
for v in 0 2 4 6; do
  LZ_DEBUG_LSTM_HOTW=$v timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4e/lab$v -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --sustain-s 0 > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/r4e/lab$v -name "*kernel_stats.csv" | head -1)
  echo "== mode $v"; python -c "
import csv,sys
for r in list(csv.DictReader(open('$f')))[:3]: print('  %-50s %8.2f us'%(r['Name'][27:77], float(r['AverageNs'])/1e3))"
done
