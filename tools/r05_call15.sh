#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "LZ_NOTHING=0" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "ROC_USE_FGS_KERNARG=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"; do echo "== bench $v"; env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain-s 0 --no-depth-sweep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
for v in "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1"; do echo "== tree_timing $v"; env $v timeout 120 python tools/tree_timing.py 2>&1 | grep -v amdgpu.ids | head -7; done
