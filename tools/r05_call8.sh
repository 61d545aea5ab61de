#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "LZ_NOTHING=0" "LZ_CHAIN_NO_SPLIT=1"; do echo "== tree_timing $v"; env $v timeout 120 python tools/tree_timing.py 2>&1 | grep -v amdgpu.ids | head -24; done
