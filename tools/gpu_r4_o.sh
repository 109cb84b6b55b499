#!/bin/bash
# r4 run o: two-pass frames with occlusion feedback, off / on, every bench workload: serial frame + four frames in flight
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
T=r4o
mkdir -p gpurun_out
for wl in cfg2 cfg3 cfg3s cfg4; do
for tp in off on; do
  timeout 400 python bench.py --workload $wl --two-pass $tp --steps 100 --warmup 30 --prewarm 60 --serial-frames 64 --no-cpu-baseline --profile-frames 1 > gpurun_out/${T}_${wl}_${tp}.json 2> gpurun_out/${T}_${wl}_${tp}.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${T}_${wl}_${tp}.json"))
    s = d["serial"]["stages_ms"]
    tp = d["config"]["two_pass"]
    print("$wl two-pass %-3s: %.0f frames/s in flight | serial %.4f ms  sort %.4f project %.4f binning %.4f composite %.4f | %s" % ("$tp", d["value"], d["serial"]["ms_per_frame"], s["sort_total"], s["project"], s["binning"], s["composite"], tp["serial_frames"]))
except Exception as e:
    print("$wl $tp failed", e)
PY
done
done
