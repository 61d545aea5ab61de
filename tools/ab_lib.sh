#!/bin/bash
# GPU A/B of two builds of the library on the same box: lightzero_amd/liblz_prev.so (a copy of an earlier build) against the current
# one, alternating; prints env-steps/s of each run
for i in 1 2; do
  for lib in lightzero_amd/liblz_prev.so lightzero_amd/liblz_mi355.so; do
    v=$(LZ_MI355_LIB=$PWD/$lib timeout 60 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; print('%.0f' % json.loads(sys.stdin.read())['value'])")
    echo "$lib $v"
  done
done
