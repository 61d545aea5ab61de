#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 0 16 4 32 20 36; do echo "== tree_timing flags $v"; LZ_DEBUG_CHAIN_FLAGS=$v timeout 120 python tools/tree_timing.py 2>&1 | grep "layer\|head conv"; done
