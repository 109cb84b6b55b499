#!/bin/bash
# band test of the cull without integer / IEEE divisions: band parity tests, then one rank's kernel durations and the band table
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "band or cfg4 or group or depth_test_second or soak" 2>&1 | tail -3
bash tools/gpu_round3_w2.sh 2>&1 | grep "msplat::" | head -12
timeout 600 python tools/band_table.py --workload cfg4 --world 8 --layouts contiguous,block:8 --out gpurun_out/r03x2_cfg4_bands.json 2>&1 | grep -v "^/opt\|    rank" | tail -4
