#!/bin/bash
# round 6: the selection by the whole workgroup (dev_select_wg; LZ_SELECT_ONE_WAVE=1 = the one-wave dev_traverse_par it replaces) -- the suites
# that replay the production launch sequence, the prologue's stamps, then a same-box A/B of the headline step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/selwg
timeout 900 python -m pytest tests/test_exact_replay_gpu.py tests/test_e2e_cfg1_gpu.py tests/test_determinism_gpu.py tests/test_shard_invariance_gpu.py tests/test_end_to_end_gpu.py tests/test_tree_gpu.py tests/test_nn_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -8
{ for v in "LZ_NOTHING=0" "LZ_SELECT_ONE_WAVE=1"; do echo "== $v"; env $v timeout 120 python tools/tree_timing.py 2>&1 | grep -v amdgpu.ids | head -7; done; } > gpurun_out/selwg/stamps.txt 2>&1
cat gpurun_out/selwg/stamps.txt
for v in "LZ_NOTHING=0" "LZ_SELECT_ONE_WAVE=1" "LZ_NOTHING=0" "LZ_SELECT_ONE_WAVE=1" "LZ_NOTHING=0" "LZ_SELECT_ONE_WAVE=1"; do
  echo "== $v"; env $v timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --sustain-s 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],4), 'sustained', round(d['config']['sustained_env_steps_per_s']), d['config']['debug_knobs'], d['config']['gpu_clock']['timed_region'], 'chain', d['roofline']['stamps']['chain_period_us'], 'lstm', d['roofline']['stamps']['lstm_period_us'])
"
done 2>&1 | tee gpurun_out/selwg/ab.txt
