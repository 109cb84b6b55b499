#!/bin/bash
# sweep: frames in flight x compositor wave pool (4 frames in flight is the bench default), then all workloads serial + default
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
run() {  # tag, bench args..., env via caller
  tag=$1; shift
  timeout 300 python bench.py --steps 400 --warmup 50 --no-cpu-baseline "$@" > gpurun_out/r2d_$tag.json 2> gpurun_out/r2d_$tag.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r2d_$tag.json").read().strip().splitlines()[-1])
    s = d["serial"]["stages_ms"] or {}
    print("%-22s fps %7.1f ms %.4f | serial %.4f ms (sort %.4f proj %.4f bin %.4f compk %.4f) | overlapped compk %.4f | hbm frac %.3f" % (
        "$tag", d["value"], d["ms_per_step"], d["serial"]["ms_per_frame"], s.get("sort_total", 0), s.get("project", 0), s.get("binning", 0),
        s.get("composite_kernel", 0), d["stages_ms"].get("composite_kernel", 0), d["frame_hbm_frac"]))
except Exception as e:
    print("$tag failed:", e); print(open("gpurun_out/r2d_$tag.err").read()[-800:])
PY
}
for P in 3 4 5 6; do
  for CW in 768 1024 1536 2048 3072; do
    MSPLAT_COMP_WAVES=$CW run p${P}_w$CW --frames-in-flight $P --serial-frames 8 --profile-frames 1
  done
done
