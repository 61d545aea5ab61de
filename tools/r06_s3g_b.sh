#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s3g_b; mkdir -p $O
timeout 900 python -m pytest tests/test_nn_golden_gpu.py tests/test_nn_gpu.py tests/test_obs64_gpu.py tests/test_go_gpu.py tests/test_muzero_gpu.py tests/test_end_to_end_gpu.py tests/test_kernel_variants_gpu.py tests/test_nn_fuzz_gpu.py tests/test_exact_replay_families_gpu.py tests/test_exact_replay_gpu.py tests/test_search_fuzz_gpu.py tests/test_sampled_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -40 > $O/pytest.log
cat $O/pytest.log
python - <<P
import json,glob
for f in sorted(glob.glob('gpurun_out/parity/*.json')):
    d=json.load(open(f))
    for k,v in d['tests'].items():
        if k.startswith('e2e/'): print(k, {a:b for a,b in v.items() if a in ('roots','identical_visit_distributions','classes','worst_gap','search_depth_max','roots_with_a_differing_selection_but_identical_visit_counts')})
P
for v in "LZ_NOTHING=0" "LZ_CHAIN_NO_SPLIT=1" "LZ_NOTHING=0"; do
  echo "== $v atari64"; env $v timeout 300 python tools/bench_conv_configs.py --family ez --obs 64 --envs 256 --sims 50 --actions 6 --steps 20 2>&1 | tail -1 | cut -c150-400
  echo "== $v go 256"; env $v timeout 300 python tools/bench_conv_configs.py --go --envs 256 --sims 200 --steps 6 --warmup 1 2>&1 | tail -1 | cut -c150-400
done 2>&1 | tee $O/ab.log
echo "== LZ_HEADS_MM64 go 256"; LZ_HEADS_MM64=1 timeout 300 python tools/bench_conv_configs.py --go --envs 256 --sims 200 --steps 6 --warmup 1 2>&1 | tail -1 | cut -c150-400
bash tools/r06_s3g_ts.sh 2>&1 | grep -v amdgpu.ids
