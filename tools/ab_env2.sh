# same-box A/B of an LZ_* switch ($1, e.g. LZ_CHAIN_WCONTIG=1) with the in-graph stamps
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for v in "LZ_NOTHING=0" "$1"; do
    env $v timeout 100 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --sustain-s 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-22s value %.0f  chain %.2f (exec %.2f)  lstm %.2f (exec %.2f)  per-sim %.2f' % ('$v', d['value'], r['avg_launch_us'], r['avg_exec_us'], r['lstm_launch_us'], r['lstm_exec_us'], r['per_simulation_us']))"
  done
done
