#!/bin/bash
# wider randomised sweeps of the network / search / tree fuzz tests (other seeds than the suite's) -> gpurun_out/parity/parity_LZ_FUZZ_SEED_OFFSET-n.json
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/parity
for off in ${SWEEP_OFFSETS:-$(seq ${SWEEP_FROM:-1} ${SWEEP_TO:-6})}; do
  echo "== offset $off"
  LZ_FUZZ_SEED_OFFSET=$off timeout 600 python -m pytest tests/test_nn_fuzz_gpu.py tests/test_search_fuzz_gpu.py tests/test_tree_fuzz_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
done
ls gpurun_out/parity
