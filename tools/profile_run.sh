#!/bin/bash
# The profiling run behind profiles/<tag>_*: GPU tests (measured parity worst cases), bench line with baselines, per-kernel stats, and the
# three PMC passes (separate runs: counters never share a run with a trace domain other than --kernel-trace).
# On the GPU box:  tools/profile_run.sh gpurun_out/<run>      then here:  python tools/refresh_profiles.py gpurun_out/<run> <tag>
R=${1:-gpurun_out/prof}
mkdir -p $R
ROOT=$(pwd)
python -c "from lightzero_amd.build import csrc_digest; import json; json.dump({'csrc_sha256': csrc_digest()}, open('$R/manifest.json', 'w'))"
rm -rf gpurun_out/parity
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $R/pytest_gpu.log 2>&1
tail -3 $R/pytest_gpu.log
mkdir -p $R/parity && cp gpurun_out/parity/*.json $R/parity/ 2>/dev/null
timeout 900 python bench.py > $R/bench_n1.json 2> $R/bench_n1.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$R/stats -- python $ROOT/bench.py --steps 15 --warmup 2 --no-cpu-baseline --sustain-s 0 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY --output-format csv -d $ROOT/$R/pmc_sq -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --sustain-s 0 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/$R/pmc_fetch -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --sustain-s 0 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ROOT/$R/pmc_write -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --sustain-s 0 > /dev/null 2>&1
cd $ROOT
# the per-dispatch traces are large: keep the summaries
find $R -name "*kernel_trace.csv" -delete
tail -c 600 $R/bench_n1.json; echo; ls -R $R | head -30
# the tree kernels' HBM traffic on BASELINE configs[2] (row n2) and the other configurations' lines, same box, same sources
bash tools/tree_traffic.sh $R/tree > $R/tree.log 2>&1
timeout 900 python tools/config_lines.py $R/cfg > $R/cfg.log 2>&1; tail -8 $R/cfg.log
timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -1
