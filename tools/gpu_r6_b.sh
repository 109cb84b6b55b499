#!/bin/bash
# round 6, run b: workgroup residency stamps (diagnostic build of the library), one frame at a time against four in flight, and with
# larger compositor pools; then the GPU suite's exchange / band tests that changed this round
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
MSPLAT_LIB_PATH=$PWD/tools/bin/variants/libmsplat_stamps.so timeout 900 python tools/stamp_timeline.py 4 48 2560 5120 > gpurun_out/r06b_stamp_timeline.txt 2>gpurun_out/r06b_err.txt
cat gpurun_out/r06b_stamp_timeline.txt | cut -c1-220; tail -3 gpurun_out/r06b_err.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "exchange or band or group" -p no:cacheprovider 2>&1 | tail -8
