#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out/r05m; mkdir -p $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $R/pytest.log 2>&1; tail -5 $R/pytest.log
for v in "LZ_NOTHING=0" "LZ_CHAIN_NO_SPLIT=1"; do
  echo "== cfg2 $v"; env $v timeout 300 python tools/bench_conv_configs.py --envs 1024 --sims 400 --steps 4 --warmup 1 2>/dev/null | tail -1 | cut -c1-300
done
echo "== cfg3"; timeout 300 python tools/bench_conv_configs.py --help 2>&1 | head -30
