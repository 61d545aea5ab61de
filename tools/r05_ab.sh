#!/bin/bash
# same-box A/B of two builds of the library: lightzero_amd/liblz_mi355_base.so (A) against the default one (B), alternating
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for lib in lightzero_amd/liblz_mi355_base.so lightzero_amd/liblz_mi355.so; do
  LZ_MI355_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --sustain-s 0 --no-depth-sweep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value']), d['ms_per_step'], d['roofline']['per_simulation_us'], round(d['fast_mode']['env_steps_per_s']))"
done; done
