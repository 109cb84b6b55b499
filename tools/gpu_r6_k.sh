#!/bin/bash
# round 6, run k: long soak and two-pass fuzz on the final tree (the sort kernels changed this round: validity in lrank, band-free pass 0)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python tools/soak.py --frames 10000 ) > gpurun_out/r06_soak_long.log 2>&1; tail -6 gpurun_out/r06_soak_long.log
( time timeout 1200 python tools/two_pass_fuzz.py --cases 400 ) > gpurun_out/r06_two_pass_fuzz.log 2>&1; tail -4 gpurun_out/r06_two_pass_fuzz.log
