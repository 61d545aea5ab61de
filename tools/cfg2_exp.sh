cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4d
timeout 600 python -m pytest tests/test_tree_gpu.py tests/test_tree_fuzz_gpu.py tests/test_exact_replay_gpu.py tests/test_muzero_gpu.py tests/test_go_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -3
for st in 1 2; do
  echo "== cfg2 streams $st"; timeout 120 python tools/bench_conv_configs.py --envs 1024 --sims 400 --steps 4 --warmup 1 --streams $st 2>/dev/null | tail -1 | cut -c150-400
done
echo "== cfg3 64"; timeout 120 python tools/bench_conv_configs.py --go --envs 64 --sims 200 --steps 6 --warmup 1 2>/dev/null | tail -1 | cut -c100-400
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4d/st -- python $GRAFT_REPO_ROOT/tools/bench_conv_configs.py --envs 1024 --sims 400 --steps 1 --warmup 1 > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r4d/st -name "*kernel_stats.csv" | head -1); python -c "
import csv
for r in list(csv.DictReader(open('$f')))[:6]: print('  %-60s %6s calls %8.2f us'%(r['Name'][27:87], r['Calls'], float(r['AverageNs'])/1e3))"
find $GRAFT_REPO_ROOT/gpurun_out/r4d/st -name "*kernel_trace.csv" -delete
