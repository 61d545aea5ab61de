# GPU A/B of two builds of the library on the same box, alternating: $1 (experimental .so, relative to the repo root) against the default one;
# prints env-steps/s and the in-graph stamps (period = first workgroup start -> next launch's first start; exec = first start -> last end)
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for lib in lightzero_amd/liblz_mi355.so $1; do
    LZ_MI355_LIB=$PWD/$lib timeout 100 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --sustain-s 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-32s value %.0f  chain %.2f (exec %.2f)  lstm %.2f (exec %.2f)  per-sim %.2f' % ('$lib'[14:], d['value'], r['avg_launch_us'], r['avg_exec_us'], r['lstm_launch_us'], r['lstm_exec_us'], r['per_simulation_us']))"
  done
done
