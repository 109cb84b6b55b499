#!/bin/bash
# r4 run r: BASELINE config 2 with the share of pass 1 pinned (two-pass frames forced on), against one pass
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
T=r4r
mkdir -p gpurun_out
WL=${1:-cfg2}
for sh in off 0.15 0.2 0.3 0.4 0.5; do
  if [ $sh = off ]; then A="--two-pass off"; else A="--two-pass on --two-pass-share $sh"; fi
  timeout 400 python bench.py --workload $WL $A --steps 200 --warmup 30 --prewarm 100 --serial-frames 64 --no-cpu-baseline --profile-frames 1 > gpurun_out/${T}_${WL}_${sh}.json 2> gpurun_out/${T}_${WL}_${sh}.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${T}_${WL}_${sh}.json"))
    s = d["serial"]["stages_ms"]
    tp = d["config"]["two_pass"]["serial_frames"]
    print("$WL share %-4s: %.0f frames/s in flight | serial %.4f ms  sort %.4f project %.4f binning %.4f composite %.4f | %s" % ("$sh", d["value"], d["serial"]["ms_per_frame"], s["sort_total"], s["project"], s["binning"], s["composite"], tp and {k: tp[k] for k in ("splats_pass1", "splats_pass2", "pairs_pass1", "pairs_pass2", "bins_unfinished")}))
except Exception as e:
    print("$WL $sh failed", e)
PY
done
