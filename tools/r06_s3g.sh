#!/bin/bash
# round 6: k_chain_s3g (split-bf16 chain on 8x8 / 9x9 / 6x7 / 4x4 grids and the GELU networks) -- parity tests, the three networks the
# round-5 sweeps left above their bound, and same-box A/B lines against the fp32 chains (LZ_CHAIN_NO_SPLIT=1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s3g; mkdir -p $O
rm -rf gpurun_out/parity
timeout 900 python -m pytest tests/test_nn_golden_gpu.py tests/test_nn_gpu.py tests/test_obs64_gpu.py tests/test_go_gpu.py tests/test_kernel_variants_gpu.py tests/test_nn_fuzz_gpu.py tests/test_exact_replay_families_gpu.py tests/test_search_fuzz_gpu.py tests/test_muzero_gpu.py tests/test_sampled_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -15 > $O/pytest.log
cat $O/pytest.log
for v in "LZ_NOTHING=0" "LZ_CHAIN_NO_SPLIT=1"; do
  env $v LZ_FUZZ_SEED_OFFSET=1447 timeout 300 python -m pytest tests/test_nn_fuzz_gpu.py -m gpu -q -p no:cacheprovider -k "configuration and 1447" 2>&1 | tail -3
  env $v LZ_FUZZ_SEED_OFFSET=600 timeout 300 python -m pytest tests/test_nn_fuzz_gpu.py -m gpu -q -p no:cacheprovider -k "sampled_model and 600" 2>&1 | tail -3
  env $v LZ_FUZZ_SEED_OFFSET=14 timeout 300 python -m pytest tests/test_nn_fuzz_gpu.py -m gpu -q -p no:cacheprovider -k "sampled_model and 14" 2>&1 | tail -3
done
python - <<P
import json,glob
for f in sorted(glob.glob('gpurun_out/parity/*.json')):
    d=json.load(open(f))
    for k,v in d['tests'].items():
        if '1447' in k or 'sez600' in k or 'sez14' in k: print(f.split('/')[-1], k, {a:b for a,b in v.items() if a in ('hc','latent','logits','policy','scalar','bounds')})
P
mkdir -p $O/parity && cp gpurun_out/parity/*.json $O/parity/
for v in "LZ_NOTHING=0" "LZ_CHAIN_NO_SPLIT=1" "LZ_NOTHING=0" "LZ_CHAIN_NO_SPLIT=1"; do
  echo "== $v atari64"; env $v timeout 300 python tools/bench_conv_configs.py --family ez --obs 64 --envs 256 --sims 50 --actions 6 --steps 20 2>&1 | tail -1 | cut -c1-400
  echo "== $v go 256"; env $v timeout 300 python tools/bench_conv_configs.py --go --envs 256 --sims 200 --steps 6 --warmup 1 2>&1 | tail -1 | cut -c1-400
done 2>&1 | tee $O/ab.log
