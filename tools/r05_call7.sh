#!/bin/bash
# round 5, GPU call 7: split-bf16 recurrent chain (k_chain_s3): numerics, exact gates, A/B
R=gpurun_out/r05g
mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_nn_golden_gpu.py tests/test_nn_gpu.py tests/test_exact_replay_gpu.py tests/test_e2e_cfg1_gpu.py tests/test_end_to_end_gpu.py tests/test_weight_refresh_gpu.py tests/test_shard_invariance_gpu.py tests/test_determinism_gpu.py -m gpu -q -p no:cacheprovider > $R/pytest.log 2>&1
tail -25 $R/pytest.log
for i in 1 2; do
for v in "LZ_CHAIN_NO_SPLIT=1" "LZ_NOTHING=0"; do
  env $v timeout 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --sustain-s 0 --no-depth-sweep 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-22s value %.0f ms %.3f  chain %.2f (exec %.2f) lstm %.2f per-sim %.2f' % ('$v', d['value'], d['ms_per_step'], r['avg_launch_us'], r['avg_exec_us'], r['lstm_launch_us'], r['per_simulation_us']))"
done; done
