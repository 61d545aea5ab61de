#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_obs64_gpu.py tests/test_nn_golden_gpu.py tests/test_go_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -5
for v in "LZ_NOTHING=0" "LZ_LSTM_NOSPLIT=1" "LZ_NOTHING=0" "LZ_LSTM_NOSPLIT=1"; do
  echo "== $v atari64"; env $v timeout 300 python tools/bench_conv_configs.py --family ez --obs 64 --envs 256 --sims 50 --actions 6 --steps 20 2>&1 | tail -1 | cut -c150-400
done 2>&1
echo "== go 256"; timeout 300 python tools/bench_conv_configs.py --go --envs 256 --sims 200 --steps 6 --warmup 1 2>&1 | tail -1 | cut -c150-400
bash tools/r06_s3g_ts.sh 2>&1 | grep -E "staged|sync|end|=="
python tools/config_lines.py gpurun_out/r06_cfgB cfg_atari64 2>&1 | tail -2
python - <<P
import json
d=json.load(open("gpurun_out/r06_cfgB/cfg_atari64.json"))
for k in d["kernels"]: print("   ", k)
P
