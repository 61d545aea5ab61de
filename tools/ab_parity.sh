# parity-mode headline with the in-graph stamps, three runs
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout 200 python bench.py --no-cpu-baseline --sustain-s 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('parity: value %.0f  ms/step %.3f  chain %.2f (exec %.2f)  lstm %.2f (exec %.2f)  per-sim %.2f' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['avg_exec_us'], r['lstm_launch_us'], r['lstm_exec_us'], r['per_simulation_us']))"
done
