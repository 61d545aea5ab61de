#!/bin/bash
# round 6: one launch per simulation (k_sim_fused, LZ_SIM_ONE_LAUNCH=1) -- the suites that replay the production launch sequence with the switch on,
# then a same-box A/B of the headline step
cd $GRAFT_REPO_ROOT
export LZ_SIM_ONE_LAUNCH=1
timeout 900 python -m pytest tests/test_exact_replay_gpu.py tests/test_e2e_cfg1_gpu.py tests/test_determinism_gpu.py tests/test_shard_invariance_gpu.py tests/test_end_to_end_gpu.py tests/test_bench_line_gpu.py tests/test_nn_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -8
unset LZ_SIM_ONE_LAUNCH
for v in "LZ_NOTHING=0" "LZ_SIM_ONE_LAUNCH=1" "LZ_NOTHING=0" "LZ_SIM_ONE_LAUNCH=1" "LZ_NOTHING=0" "LZ_SIM_ONE_LAUNCH=1"; do
  echo "== $v"; env $v timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --sustain-s 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],4), 'sustained', round(d['config']['sustained_env_steps_per_s']), d['config']['debug_knobs'], d['config']['gpu_clock']['timed_region'])
"
done
