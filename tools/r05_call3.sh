#!/bin/bash
# round 5, GPU call 3: the XCD-local resident hand-off ubench, the refresh-every-step bench arm, the whole-host CPU arm
R=gpurun_out/r05c
mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 300 python tools/ubench/xcd_resident.py $R/xcd_resident.json 2>&1 | tee $R/xcd_resident.log | tail -12
timeout 120 python tools/ubench/handoff.py 2>&1 | tail -8 | tee $R/handoff.log
timeout 300 python -m pytest tests/test_weight_refresh_gpu.py -q -p no:cacheprovider 2>&1 | tail -3
for a in "" "--refresh-every 1" "--refresh-every 4"; do
timeout 300 python bench.py $a --no-cpu-baseline --sustain-s 0 --no-depth-sweep --steps 40 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('bench $a: value %.0f ms %.3f' % (d['value'], d['ms_per_step']))"
done
timeout 400 python - <<'P'
import bench, json
r = bench.cpu_baseline_whole_host(16, 0)
print(json.dumps({k: v for k, v in r.items() if k != 'sample'})[:1500])
P
