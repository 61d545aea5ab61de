#!/bin/bash
# round 6: MuZero's head MLPs in the tail of the chain launch (k_chain_s3_th; LZ_NO_TAIL_HEADS=1 = the separate head launch) -- the MuZero suites,
# then same-box A/B on MuZero Atari 96x96 at 256 roots and on BASELINE configs[2] (1024 x 400)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tailheads
timeout 900 python -m pytest tests/test_muzero_gpu.py tests/test_nn_golden_gpu.py tests/test_nn_gpu.py tests/test_exact_replay_gpu.py tests/test_determinism_gpu.py tests/test_reanalyze_gpu.py tests/test_gumbel_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -6
for v in "LZ_NOTHING=0" "LZ_NO_TAIL_HEADS=1" "LZ_NOTHING=0" "LZ_NO_TAIL_HEADS=1"; do
  echo "== $v"
  env $v timeout 600 python tools/bench_conv_configs.py --envs 256 --sims 50 --actions 6 --steps 20 --warmup 3 2>&1 | grep -v amdgpu.ids | tail -1
  env $v timeout 600 python tools/bench_conv_configs.py --envs 512 --sims 50 --actions 6 --steps 10 --warmup 2 2>&1 | grep -v amdgpu.ids | tail -1
done 2>&1 | tee gpurun_out/tailheads/ab.txt
