#!/bin/bash
# Row n2 of the coverage table: rocprof HBM traffic of the backup / selection kernels at HEAD, on the configuration where the tree step is
# a launch of its own -- BASELINE configs[2] (Atari MuZero, 1024 roots x 400 simulations, A = 4): once a root's tree outgrows the LDS
# budget the step runs as k_backprop_traverse<1, 1> on the HBM node arrays.  Three separate rocprofv3 passes (kernel trace + stats;
# --pmc FETCH_SIZE; --pmc WRITE_SIZE -- counters never share a run with another trace domain), then tools/tree_traffic.py summarises.
#   on the GPU box:  tools/tree_traffic.sh gpurun_out/<run>      here:  python tools/tree_traffic.py gpurun_out/<run> r04
R=$1; case $R in /*) ;; *) R=$GRAFT_REPO_ROOT/${R:-gpurun_out/tree};; esac; mkdir -p $R
CMD="python $GRAFT_REPO_ROOT/tools/bench_conv_configs.py --envs 1024 --sims 400 --steps 1 --warmup 1"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats -- $CMD > $R/line.json 2> /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/pmc_fetch -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/pmc_write -- $CMD > /dev/null 2>&1
find $R -name "*kernel_trace.csv" -delete
python -c "from lightzero_amd.build import csrc_digest; import json; json.dump({'csrc_sha256': csrc_digest()}, open('$R/manifest.json', 'w'))" 2>/dev/null || (cd $GRAFT_REPO_ROOT && python -c "from lightzero_amd.build import csrc_digest; import json; json.dump({'csrc_sha256': csrc_digest()}, open('$R/manifest.json', 'w'))")
ls -R $R | head -20
