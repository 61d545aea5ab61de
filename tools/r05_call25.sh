#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== tree_timing"; timeout 120 python tools/tree_timing.py 2>&1 | grep -v amdgpu.ids | head -16
timeout 600 python -m pytest tests/test_nn_golden_gpu.py tests/test_nn_gpu.py tests/test_fast_mode_gpu.py tests/test_exact_replay_gpu.py -q 2>&1 | tail -3
for v in "LZ_NOTHING=0" "LZ_NOTHING=1"; do echo "== bench $v"; env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain-s 0 --no-depth-sweep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['per_simulation_us'], d['fast_mode']['env_steps_per_s'])"; done
