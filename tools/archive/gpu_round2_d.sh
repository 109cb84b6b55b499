#!/bin/bash
# frames in flight x hardware queues
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
run() {
  tag=$1; shift
  timeout 300 python bench.py --steps 500 --warmup 50 --no-cpu-baseline --serial-frames 8 --profile-frames 1 "$@" > gpurun_out/r2d_$tag.json 2> gpurun_out/r2d_$tag.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r2d_$tag.json").read().strip().splitlines()[-1])
    print("%-22s fps %7.1f ms %.4f | overlapped compk %.4f | hbm frac %.3f | enqueue %.4f" % ("$tag", d["value"], d["ms_per_step"], d["stages_ms"].get("composite_kernel", 0), d["frame_hbm_frac"], d["host_enqueue_ms_per_frame"]))
except Exception as e:
    print("$tag failed:", e); print(open("gpurun_out/r2d_$tag.err").read()[-800:])
PY
}
for Q in 4 6 8 12 16 24; do
  GPU_MAX_HW_QUEUES=$Q run p4_q$Q --frames-in-flight 4
done
GPU_MAX_HW_QUEUES=12 run p5_q12 --frames-in-flight 5
GPU_MAX_HW_QUEUES=12 run p6_q12 --frames-in-flight 6
GPU_MAX_HW_QUEUES=24 run p6_q24 --frames-in-flight 6
