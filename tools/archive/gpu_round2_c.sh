#!/bin/bash
# PMC counters of the compositor (one pass per counter group) + probe dump
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $grp | cut -c1-10 | tr ' ' '_')
  bash tools/gpu_pmc.sh r2c_$tag "$grp" --frames-in-flight 1 --prewarm 20 --serial-frames 8 2>&1 | grep -E "kernel |composite|project" | cut -c1-260
done
timeout 200 python tools/probe_dump.py cfg2 0 2>&1 | grep -E "step|work"
