#!/bin/bash
# round 6, run s: with two frames per half of the CUs, do the in-flight kernel choices still hold?
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
p() { timeout 200 python tools/cu_partition_probe.py "$@" 2>>gpurun_out/r06s_err.txt; }
for rep in 1 2; do
  p --label "shared"                    --parts 0,0,0,0
  p --label "halves"                    --parts 1,2,1,2
  p --label "halves, serial kernels"    --parts 1,2,1,2 --frame-mode serial
  MSPLAT_SORT=lsd8 p --label "halves, 8-bit sort passes" --parts 1,2,1,2
  p --label "three on all + ..."        --parts 1,2,0,0
  p --label "halves, 6 in flight"       --parts 1,2,1,2,1,2
  p --label "halves, 8 in flight"       --parts 1,2,1,2,1,2,1,2
  p --label "halves, 2 in flight"       --parts 1,2
done
tail -3 gpurun_out/r06s_err.txt
