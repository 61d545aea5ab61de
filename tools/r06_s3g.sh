#!/bin/bash
# round 6: k_chain_s3g (split-bf16 chain on 8x8 / 9x9 / 6x7 / 4x4 grids and the GELU networks) -- the three networks the round-5 sweeps
# left above their bound under both chains (-> profiles/r06_named_networks.json), and same-box A/B lines against the fp32 chains
# (LZ_CHAIN_NO_SPLIT=1), the head launch (LZ_HEADS_LAUNCH=1) and the 99 KB LSTM (LZ_LSTM_NO_OVL=1) -> profiles/r06_s3g_ab.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s3g; mkdir -p $O
rm -rf gpurun_out/parity
for v in "LZ_NOTHING=0" "LZ_CHAIN_NO_SPLIT=1"; do
  env $v timeout 300 python -m pytest tests/test_nn_fuzz_gpu.py -m gpu -q -p no:cacheprovider -k "named" 2>&1 | tail -2
done
python - <<P
import json,glob
out={}
for f in sorted(glob.glob('gpurun_out/parity/*.json')):
    d=json.load(open(f))
    for k,v in d['tests'].items():
        if k.startswith('named/'): out.setdefault(k, {})[d['variant']] = v
json.dump({"what": "the three networks of profiles/r05_parity_sweeps.json 'above_their_bound', as named cases (tests/test_nn_fuzz_gpu.py), under the "
           "split-bf16 chain k_chain_s3g (variant LZ_NOTHING=0 = default) and the fp32 chains it replaced (LZ_CHAIN_NO_SPLIT=1); unit |device - torch fp32| / (1 + |x|); "
           "'bounds' = max(north_star's, 3 x torch fp32's own distance from binary64 on that network)", "tests": out}, open('$O/named_networks.json','w'), indent=1, sort_keys=True)
for k,v in out.items():
    for var,e in v.items(): print(k, var, {a:b for a,b in e.items() if a in ('hc','latent','logits','policy','scalar')})
P
for v in "LZ_NOTHING=0" "LZ_CHAIN_NO_SPLIT=1" "LZ_HEADS_LAUNCH=1" "LZ_LSTM_NO_OVL=1" "LZ_NOTHING=0" "LZ_CHAIN_NO_SPLIT=1" "LZ_HEADS_LAUNCH=1" "LZ_LSTM_NO_OVL=1"; do
  echo "== $v atari64"; env $v timeout 300 python tools/bench_conv_configs.py --family ez --obs 64 --envs 256 --sims 50 --actions 6 --steps 20 2>&1 | tail -1 | cut -c1-400
done 2>&1 | tee $O/ab.txt
for v in "LZ_NOTHING=0" "LZ_CHAIN_NO_SPLIT=1" "LZ_HEADS_MM64=1" "LZ_NOTHING=0" "LZ_CHAIN_NO_SPLIT=1" "LZ_HEADS_MM64=1"; do
  echo "== $v go 256"; env $v timeout 300 python tools/bench_conv_configs.py --go --envs 256 --sims 200 --steps 6 --warmup 1 2>&1 | tail -1 | cut -c1-400
done 2>&1 | tee -a $O/ab.txt
