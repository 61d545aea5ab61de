#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05i
timeout 900 python -m pytest tests/test_nn_golden_gpu.py tests/test_nn_gpu.py tests/test_exact_replay_gpu.py tests/test_e2e_cfg1_gpu.py tests/test_determinism_gpu.py -q 2>&1 | tail -8
for v in "LZ_NOTHING=0" "LZ_CHAIN_NO_SPLIT=1"; do echo "== bench $v"; env $v timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
echo "== tree_timing"; timeout 120 python tools/tree_timing.py 2>&1 | grep -v amdgpu.ids | head -24
