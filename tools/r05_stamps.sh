#!/bin/bash
# phase stamps of the fused chain launch (debug build of the library, same sources): profiles/r05_chain_stamps.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05stamps
{ echo "# tools/tree_timing.py (liblz_mi355_dbg.so, LZ_NO_GRAPH=1): s_memtime cycle stamps of root 0's workgroup in the last tree-fused chain launch"
  echo "# of a 256-root x 50-simulation search, EfficientZero Atari model; three runs"
  for i in 1 2 3; do echo "== run $i"; timeout 120 python tools/tree_timing.py 2>&1 | grep -v amdgpu.ids; done
  echo "# LZ_CHAIN_NO_SPLIT=1 (k_chain_w, the fp32 Winograd chain the split-bf16 chain replaced)"
  LZ_CHAIN_NO_SPLIT=1 timeout 120 python tools/tree_timing.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r05stamps/chain_stamps.txt
tail -30 gpurun_out/r05stamps/chain_stamps.txt
