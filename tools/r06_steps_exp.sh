#!/bin/bash
# round 6: what a timed region of K steps measures -- the interpreter's generation-2 collection (30-50 ms of host time) lands in some regions and not in
# others; bench.py freezes the heap it has built before the warm-up (gc.freeze(); LZ_BENCH_NO_GC_FREEZE=1 = before), LZ_BENCH_NO_CLOCK=1 = no clock sampler thread
cd $GRAFT_REPO_ROOT
run() { env $1 $3 timeout 300 python bench.py --steps $2 --warmup 5 --no-cpu-baseline --no-depth-sweep --sustain-s 0.5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 $3 steps $2:', round(d['value']), round(d['ms_per_step'],4), 'sustained', round(d['config']['sustained_env_steps_per_s']), (d['config']['gpu_clock'] or {}).get('timed_region'))
"; }
for k in 20 40 80 20 40 80; do run LZ_X=0 $k; done
for k in 20 40 80; do run LZ_BENCH_NO_GC_FREEZE=1 $k; done
for k in 20 40 80; do run LZ_BENCH_NO_GC_FREEZE=1 $k LZ_BENCH_NO_CLOCK=1; done
