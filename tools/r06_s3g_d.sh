#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_obs64_gpu.py tests/test_nn_golden_gpu.py tests/test_nn_gpu.py tests/test_exact_replay_families_gpu.py tests/test_exact_replay_gpu.py tests/test_search_fuzz_gpu.py tests/test_kernel_variants_gpu.py tests/test_determinism_gpu.py tests/test_shard_invariance_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -15
for v in "LZ_NOTHING=0" "LZ_HEADS_LAUNCH=1" "LZ_NOTHING=0" "LZ_HEADS_LAUNCH=1"; do
  echo "== $v atari64"; env $v timeout 300 python tools/bench_conv_configs.py --family ez --obs 64 --envs 256 --sims 50 --actions 6 --steps 20 2>&1 | tail -1 | cut -c150-400
done 2>&1
