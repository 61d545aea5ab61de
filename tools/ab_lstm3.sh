cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4e
python tools/dump_search.py gpurun_out/r4e/a.npz 256 50 2>/dev/null | tail -1
LZ_LSTM3=1 python tools/dump_search.py gpurun_out/r4e/b.npz 256 50 2>/dev/null | tail -1
python tools/dump_search.py --compare gpurun_out/r4e/a.npz gpurun_out/r4e/b.npz
python tools/dump_search.py gpurun_out/r4e/a2.npz 67 12 2>/dev/null | tail -1
LZ_LSTM3=1 python tools/dump_search.py gpurun_out/r4e/b2.npz 67 12 2>/dev/null | tail -1
python tools/dump_search.py --compare gpurun_out/r4e/a2.npz gpurun_out/r4e/b2.npz
rm -f gpurun_out/r4e/*.npz
for i in 1 2; do
  for v in "" "LZ_LSTM3=1"; do
    env $v timeout 100 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --sustain-s 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', 'value %.0f  chain %.2f us  lstm %.2f us (exec %.2f)  per-sim %.2f' % (d['value'], r['avg_launch_us'], r['lstm_launch_us'], r['lstm_exec_us'], r['per_simulation_us']))"
  done
done
