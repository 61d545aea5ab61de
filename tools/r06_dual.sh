#!/bin/bash
# round 6: the downsample block's two stride-2 convolutions in one launch (k_conv_s3<32, 64, 2, true>; LZ_CONV_NO_DUAL=1 = two launches) and
# lz_model_cfg.precision = 2 -- goldens / teacher-forced networks, then a same-box A/B of the headline step and the tower's per-kernel durations
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dual
timeout 900 python -m pytest tests/test_nn_golden_gpu.py tests/test_nn_gpu.py tests/test_obs64_gpu.py tests/test_e2e_cfg1_gpu.py tests/test_reference_forward_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -8
for v in "LZ_NOTHING=0" "LZ_CONV_NO_DUAL=1" "LZ_NOTHING=0" "LZ_CONV_NO_DUAL=1" "LZ_NOTHING=0" "LZ_CONV_NO_DUAL=1"; do
  echo "== $v"; env $v timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --sustain-s 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],4), 'sustained', round(d['config']['sustained_env_steps_per_s']), d['config']['debug_knobs'], d['config']['gpu_clock']['timed_region'])
"
done 2>&1 | tee gpurun_out/dual/ab.txt
bash tools/tower_trace.sh 2>&1 | tee gpurun_out/dual/tower.txt
LZ_CONV_NO_DUAL=1 bash tools/tower_trace.sh 2>&1 | tee gpurun_out/dual/tower_nodual.txt
