#!/bin/bash
# round 5, GPU call 1: the tree-parallel selection (dev_traverse_par) -- parity first, then the bench line with the depth sweep, then the same-box A/B
R=gpurun_out/r05a
mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_tree_gpu.py tests/test_end_to_end_gpu.py tests/test_exact_replay_gpu.py tests/test_e2e_cfg1_gpu.py tests/test_tree_fuzz_gpu.py \
    tests/test_search_fuzz_gpu.py tests/test_checkpoint_gpu.py tests/test_mlp_models_gpu.py tests/test_obs64_gpu.py tests/test_shard_invariance_gpu.py \
    -m gpu -q -x -p no:cacheprovider > $R/pytest.log 2>&1
tail -15 $R/pytest.log
timeout 600 python bench.py > $R/bench.json 2> $R/bench.err
tail -c 1500 $R/bench.json; echo
for i in 1 2; do
  for v in "LZ_TRAVERSE_SERIAL=1" "LZ_NOTHING=0"; do
    env $v timeout 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --sustain-s 0 2>/dev/null | tail -1 > $R/ab_${v}_$i.json
    python - $R/ab_${v}_$i.json "$v" <<'P'
import sys,json
d=json.loads(open(sys.argv[1]).read()); r=d['roofline']; c=d['config']
print('%-22s value %.0f  chain %.2f (exec %.2f)  lstm %.2f  per-sim %.2f  depth %.2f/%d' % (sys.argv[2], d['value'], r['avg_launch_us'], r['avg_exec_us'], r['lstm_launch_us'], r['per_simulation_us'], c['search_depth_mean'], c['search_depth_max']))
for k,a in c.get('depth_sweep',{}).get('arms',{}).items():
    print('    scale %-3s %s' % (k, {x:(round(y,2) if isinstance(y,float) else y) for x,y in a.items() if x in ('env_steps_per_s','search_depth_mean','search_depth_max','chain_period_us','vs_scale_1','root_prior_max_prob_mean','error')}))
P
  done
done
for v in "LZ_TRAVERSE_SERIAL=1" "LZ_NOTHING=0"; do echo "== tree_timing $v"; env $v timeout 120 python tools/tree_timing.py 2>&1 | head -8; done
