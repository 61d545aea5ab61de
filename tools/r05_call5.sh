#!/bin/bash
R=gpurun_out/r05e
mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 300 python tools/refresh_probe.py 2>&1 | tail -6
echo "== cfg2 default"; timeout 300 python tools/bench_conv_configs.py --envs 1024 --sims 400 --steps 4 --warmup 1 2>/dev/null | tail -1 | cut -c150-330
timeout 400 python - <<'P'
import bench, json
r = bench.cpu_baseline_whole_host(16, 0)
print(json.dumps({k: v for k, v in r.items() if k != 'sample'})[:1200])
P
timeout 600 python -m pytest tests/test_exact_replay_gpu.py tests/test_end_to_end_gpu.py tests/test_bench_multirank_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
