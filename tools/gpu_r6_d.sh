#!/bin/bash
# round 6, run d: the stamps again, grouped into launches (how far apart do a launch's workgroups START?), plus a slice for offline viewing
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
MSPLAT_LIB_PATH=$PWD/tools/bin/variants/libmsplat_stamps.so timeout 900 python tools/stamp_timeline.py 4 96 > gpurun_out/r06d_stamp_timeline.txt 2>gpurun_out/r06d_err.txt
cat gpurun_out/r06d_stamp_timeline.txt | cut -c1-220; tail -3 gpurun_out/r06d_err.txt
