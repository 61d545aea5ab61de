#!/bin/bash
# GPU A/B of environment knobs: NN golden tests + bench line for the default and for each "VAR=1" argument; with PROF=1 also per-kernel stats
mkdir -p gpurun_out
for v in "" "$@"; do
  tag=${v:-default}
  (env $v timeout 90 python -m pytest tests/test_nn_golden_gpu.py tests/test_nn_gpu.py -x -q 2>&1 | tail -2) > gpurun_out/abe_t_$tag.log
  (env $v timeout 60 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-230) > gpurun_out/abe_b_$tag.log
  echo "== $tag"; cat gpurun_out/abe_t_$tag.log gpurun_out/abe_b_$tag.log
  if [ -n "$PROF" ]; then
    (cd /tmp && export TMPDIR=/tmp && env $v timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abe_prof_$tag -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1)
    f=$(find gpurun_out/abe_prof_$tag -name "*kernel_stats.csv" | head -1); head -12 $f | cut -d, -f1-4,6,7 | cut -c1-150
  fi
done
