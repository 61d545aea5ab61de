# debug build: where a layer of k_chain_b spends its cycles (LZ_DEBUG_CHAIN_FLAGS: 4 no pixel reads / MFMAs, 8 no epilogue, 16 no weight stream)
cd $GRAFT_REPO_ROOT
for f in 0 4 8 16 12 28; do
  echo "== LZ_DEBUG_CHAIN_FLAGS=$f"
  LZ_DEBUG_CHAIN_FLAGS=$f LZ_TOOL_FAST=1 timeout 200 python tools/tree_timing.py 2>&1 | grep "layer\|head conv\|latent"
done
