#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "LZ_NOTHING=0" "LZ_CHAIN_NO_SPLIT=1"; do env $v timeout 600 python -m pytest tests/test_go_gpu.py -m gpu -q -p no:cacheprovider -s 2>&1 | grep -E "go worst|passed|failed|Error" ; done
for v in "LZ_NOTHING=0" "LZ_NOTHING=0"; do
  echo "== $v atari64"; env $v timeout 300 python tools/bench_conv_configs.py --family ez --obs 64 --envs 256 --sims 50 --actions 6 --steps 20 2>&1 | tail -1 | cut -c150-400
  echo "== $v go 256"; env $v timeout 300 python tools/bench_conv_configs.py --go --envs 256 --sims 200 --steps 6 --warmup 1 2>&1 | tail -1 | cut -c150-400
done 2>&1
bash tools/r06_s3g_ts.sh 2>&1 | grep -E "staged|sync|end|==|L2|layer"
