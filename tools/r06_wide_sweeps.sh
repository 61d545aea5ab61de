#!/bin/bash
# round 6: the randomised tree / search fuzz at other seeds than the suite's with every MuZero / EfficientZero tree step on the chunk-loop
# kernels of lz_tree_wide.hip (LZ_TREE_WIDE=1; LZ_NO_TREE_FUSE=1 makes the fused searches launch their tree steps separately, i.e. through them)
cd $GRAFT_REPO_ROOT
for off in $(seq ${SWEEP_FROM:-1} ${SWEEP_TO:-30}); do
  echo "== offset $off"
  LZ_TREE_WIDE=1 LZ_NO_TREE_FUSE=1 LZ_FUZZ_SEED_OFFSET=$off timeout 600 python -m pytest tests/test_tree_fuzz_gpu.py tests/test_search_fuzz_gpu.py tests/test_tree_wide_gpu.py -k "random" -m gpu -q -p no:cacheprovider 2>&1 | tail -2
done
