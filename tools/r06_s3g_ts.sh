#!/bin/bash
# cycle stamps of k_chain_s3g's workgroup 0 (debug build, no graph, tree step in its own launch)
cd $GRAFT_REPO_ROOT
export LZ_MI355_LIB=$PWD/lightzero_amd/liblz_mi355_dbg.so LZ_DEBUG_CHAIN_TS=1 LZ_NO_GRAPH=1
echo "== atari64 (8x8)"; timeout 300 python tools/bench_conv_configs.py --family ez --obs 64 --envs 256 --sims 50 --actions 6 --steps 2 --warmup 1 --chain-ts 2>&1 | tail -30
echo "== go 256 (9x9)"; timeout 300 python tools/bench_conv_configs.py --go --envs 256 --sims 200 --steps 1 --warmup 1 --chain-ts 2>&1 | tail -30
