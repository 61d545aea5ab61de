#!/bin/bash
# GPU A/B of the chain variants: NN golden tests, bench line and phase stamps for the default and for each "VAR=1" given as arguments
mkdir -p gpurun_out
(timeout 300 python -m pytest tests/test_nn_golden_gpu.py tests/test_nn_gpu.py -x -q 2>&1 | tail -4) > gpurun_out/ab_t.log
for v in "" "$@"; do
  tag=${v:-default}
  (env $v timeout 60 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-230) > gpurun_out/ab_b_$tag.log
  (env $v timeout 60 python tools/chain_timing.py 2>&1 | tail -22) > gpurun_out/ab_ts_$tag.log
  if [ -n "$v" ]; then (env $v timeout 120 python -m pytest tests/test_nn_golden_gpu.py -x -q 2>&1 | tail -2) > gpurun_out/ab_t_$tag.log; fi
done
cat gpurun_out/ab_t*.log
for v in "" "$@"; do tag=${v:-default}; echo "== $tag"; cat gpurun_out/ab_b_$tag.log; done
f=""; for v in "" "$@"; do f="$f gpurun_out/ab_ts_${v:-default}.log"; done
paste $f | cut -c1-60,75-120,150-200
