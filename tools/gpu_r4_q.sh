#!/bin/bash
# r4 run q: two-pass AUTO vs off under the driver's protocol and in long blocks
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
T=r4q
mkdir -p gpurun_out
for wl in cfg2 cfg3 cfg3s cfg4 cfg5; do
for tp in off auto; do
  timeout 400 python bench.py --workload $wl --two-pass $tp --steps 20 --warmup 5 --no-cpu-baseline --profile-frames 1 > gpurun_out/${T}_${wl}_${tp}.json 2> gpurun_out/${T}_${wl}_${tp}.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${T}_${wl}_${tp}.json"))
    s = d["serial"]["stages_ms"]
    tp = d["config"]["two_pass"]
    print("$wl two-pass %-4s: %.0f frames/s (20-frame blocks) | serial %.4f ms  sort %.4f project %.4f binning %.4f composite %.4f | flight %s | serial %s" % ("$tp", d["value"], d["serial"]["ms_per_frame"], s["sort_total"], s["project"], s["binning"], s["composite"], tp["timed_region"] and {k: tp["timed_region"][k] for k in ("frames_total", "share_pass1", "bins_unfinished")}, tp["serial_frames"] and {k: tp["serial_frames"][k] for k in ("frames", "share_pass1", "bins_unfinished")}))
except Exception as e:
    print("$wl $tp failed", e)
PY
done
done
