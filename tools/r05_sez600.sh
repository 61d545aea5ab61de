#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "LZ_NOTHING=0" "LZ_CONV_NO_SPLIT=1 LZ_CHAIN_NO_SPLIT=1"; do
  rm -rf gpurun_out/parity
  env $v LZ_FUZZ_SEED_OFFSET=600 timeout 300 python -m pytest tests/test_nn_fuzz_gpu.py -m gpu -q -p no:cacheprovider -k "sez or conv_sampled" 2>&1 | tail -2
  python - <<P
import json,glob
for f in glob.glob('gpurun_out/parity/*.json'):
    d=json.load(open(f))
    for k,v in d['tests'].items():
        if 'sez600' in k or 'sez14' in k: print("$v", f.split('/')[-1], k, {a:b for a,b in v.items() if a in ('hc','latent','logits','policy','scalar')})
P
done
