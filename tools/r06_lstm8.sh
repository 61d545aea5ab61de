#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_vector_collector_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
for v in "LZ_NOTHING=0" "LZ_LSTM_NO_OVL=1" "LZ_NOTHING=0" "LZ_LSTM_NO_OVL=1" "LZ_NOTHING=0" "LZ_LSTM_NO_OVL=1"; do
  echo "== $v atari64"; env $v timeout 300 python tools/bench_conv_configs.py --family ez --obs 64 --envs 256 --sims 50 --actions 6 --steps 30 2>&1 | tail -1 | cut -c150-400
done
