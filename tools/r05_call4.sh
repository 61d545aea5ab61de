#!/bin/bash
# round 5, GPU call 4: workgroup tree step for deep trees (configs[2]), XCD-local resident ubench, refresh-in-the-loop timing, CPU allowance of the box
R=gpurun_out/r05d
mkdir -p $R
cd $GRAFT_REPO_ROOT
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null) ; cfs_quota: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) ; nproc $(nproc)"
timeout 900 python -m pytest tests/test_end_to_end_gpu.py tests/test_weight_refresh_gpu.py "tests/test_exact_replay_gpu.py::test_configs2_deep_tree_muzero_full_size_replays_exactly" tests/test_search_fuzz_gpu.py tests/test_tree_fuzz_gpu.py -m gpu -q -p no:cacheprovider > $R/pytest.log 2>&1
tail -12 $R/pytest.log
timeout 300 python tools/ubench/xcd_resident.py $R/xcd_resident.json 2>&1 | tee $R/xcd_resident.log | tail -10
for v in "LZ_TREE_NO_WG=1" "LZ_NOTHING=0" "LZ_TREE_NO_WG=1" "LZ_NOTHING=0"; do
  echo "== cfg2 $v"; env $v timeout 300 python tools/bench_conv_configs.py --envs 1024 --sims 400 --steps 4 --warmup 1 2>/dev/null | tail -1 | cut -c1-300
done
for v in "LZ_TREE_LDS_LIMIT=8192" "LZ_TREE_LDS_LIMIT=0"; do
  echo "== cfg2 $v"; env $v timeout 300 python tools/bench_conv_configs.py --envs 1024 --sims 400 --steps 4 --warmup 1 2>/dev/null | tail -1 | cut -c1-300
done
for a in "" "--refresh-every 1" "--refresh-every 4"; do
timeout 300 python bench.py $a --no-cpu-baseline --sustain-s 0 --no-depth-sweep --steps 40 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('bench $a: value %.0f ms %.3f' % (d['value'], d['ms_per_step']))"
done
