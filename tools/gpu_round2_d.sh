#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
run() {
  tag=$1; shift
  timeout 300 python bench.py --steps 500 --warmup 50 --no-cpu-baseline --serial-frames 8 --profile-frames 1 "$@" > gpurun_out/r2d_$tag.json 2> gpurun_out/r2d_$tag.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r2d_$tag.json").read().strip().splitlines()[-1])
    print("%-22s fps %7.1f ms %.4f | overlapped compk %.4f | hbm frac %.3f | blocks %s" % ("$tag", d["value"], d["ms_per_step"], d["stages_ms"].get("composite_kernel", 0), d["frame_hbm_frac"], [round(b,1) for b in d["block_ms"]]))
except Exception as e:
    print("$tag failed:", e); print(open("gpurun_out/r2d_$tag.err").read()[-800:])
PY
}
for P in 3 4 5; do
  for CW in 1024 1536 2048 3072 4096; do
    MSPLAT_COMP_WAVES=$CW run p${P}_w$CW --frames-in-flight $P
  done
done
GPU_MAX_HW_QUEUES=16 MSPLAT_COMP_WAVES=1536 run p6_q16_w1536 --frames-in-flight 6
GPU_MAX_HW_QUEUES=16 MSPLAT_COMP_WAVES=1536 run p8_q16_w1536 --frames-in-flight 8
