#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== tree_timing"; timeout 120 python tools/tree_timing.py 2>&1 | grep -v amdgpu.ids | grep layer
