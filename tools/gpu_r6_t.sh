#!/bin/bash
# round 6, run t: under the CU halves, does the projection with 5 KB of LDS (tools/gpu_r6_n.sh: -3 % on shared CUs) change sign?
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
p() { timeout 200 python tools/cu_partition_probe.py "$@" 2>>gpurun_out/r06t_err.txt; }
V=$PWD/tools/bin/variants
for rep in 1 2 3; do
  p --label "halves, product"                    --parts 1,2,1,2
  MSPLAT_LIB_PATH=$V/libmsplat_parts4.so p --label "halves, projection 5 KB LDS"   --parts 1,2,1,2
done
MSPLAT_LIB_PATH=$V/libmsplat_parts4.so p --label "shared, projection 5 KB LDS"   --parts 0,0,0,0
tail -3 gpurun_out/r06t_err.txt
