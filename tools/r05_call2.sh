#!/bin/bash
# round 5, GPU call 2: full GPU suite (tree-parallel selection, device-side weight refresh, split collector forward), then the bench line
R=gpurun_out/r05b
mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $R/pytest.log 2>&1
tail -40 $R/pytest.log
timeout 900 python bench.py > $R/bench.json 2> $R/bench.err
tail -3 $R/bench.err
python - $R/bench.json <<'P'
import sys,json
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print('value %.0f  ms %.3f  chain %.2f lstm %.2f per-sim %.2f frac %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['lstm_launch_us'], r['per_simulation_us'], r['frac']))
for k in ('search_depth_mean','search_depth_max','weight_refresh_ms','weight_refresh_device_ms','collector_env_steps_per_s','collector_2groups_env_steps_per_s','policy_surface_env_steps_per_s','sustained_env_steps_per_s','speedup_vs_cpu_baseline','speedup_vs_cpu_baseline_whole_host'):
    print(' ', k, c.get(k))
for k,a in c.get('depth_sweep',{}).get('arms',{}).items():
    print('    scale %-3s %s' % (k, {x:(round(y,2) if isinstance(y,float) else y) for x,y in a.items() if x in ('env_steps_per_s','search_depth_mean','search_depth_max','chain_period_us','vs_scale_1','error')}))
print(' cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['thread_sweep_env_steps_per_s'])
w=d.get('cpu_baseline_whole_host',{}); print(' whole', w.get('value'), [(x.get('processes'),x.get('threads_per_process'),round(x.get('value',0),1),x.get('error')) for x in w.get('configurations',[])], w.get('error'))
print(' fast', d['fast_mode'].get('env_steps_per_s'), d['fast_mode'].get('error'))
P
timeout 300 python bench.py --refresh-every 1 --no-cpu-baseline --sustain-s 0 --no-depth-sweep 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('refresh-every-1: value %.0f ms %.3f' % (d['value'], d['ms_per_step']))"
