# both modes with the in-graph stamps, two runs each, then the phase stamps of the fused chain launch (debug library)
cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout 200 python bench.py --no-cpu-baseline --sustain-s 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; f=d.get('fast_mode') or {}; print('parity: value %.0f  ms/step %.3f  chain %.2f (exec %.2f)  lstm %.2f (exec %.2f)  per-sim %.2f   | fast arm %s' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['avg_exec_us'], r['lstm_launch_us'], r['lstm_exec_us'], r['per_simulation_us'], f.get('env_steps_per_s', f)))"
timeout 200 python bench.py --fast --no-cpu-baseline --sustain-s 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('fast:   value %.0f  ms/step %.3f  chain %.2f (exec %.2f)  lstm %.2f (exec %.2f)  per-sim %.2f' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['avg_exec_us'], r['lstm_launch_us'], r['lstm_exec_us'], r['per_simulation_us']))"
done
timeout 200 python tools/tree_timing.py 2>&1 | grep -v amdgpu.ids | head -8
