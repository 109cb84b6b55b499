#!/bin/bash
# pass 0 of the sort compacting the survivors (few visible splats): parity, soak, one band rank's kernels, band table, cfg3s
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03y2
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "band or cfg4 or group or large_cloud or scene or soak or sort or in_flight" 2>&1 | tail -3
MSPLAT_WS_COMPACT=2 timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "sort or image_matches or frame_modes or in_flight or cfg2 or wide" 2>&1 | tail -3
timeout 300 python tools/soak.py --frames 4000 2>&1 | tail -2
bash tools/gpu_round3_w2.sh 2>&1 | grep "msplat::" | head -12
timeout 600 python tools/band_table.py --workload cfg4 --world 8 --layouts contiguous,block:8 --out gpurun_out/${T}_cfg4_bands.json 2>&1 | grep -v "^/opt\|    rank" | tail -4
one() {  # name, env..., -- bench args
  name=$1; shift
  envs=(X=1)
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py --no-cpu-baseline --steps 100 --warmup 20 --prewarm 50 --profile-frames 2 "$@" > gpurun_out/${T}_$name.json 2> gpurun_out/${T}_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${T}_$name.json").read().strip().splitlines()[-1])
    s = d["serial"]["stages_ms"]
    print("%-22s fps %.0f  serial %.4f ms  sort %.4f  project %.4f  binning %.4f  comp %.4f" % ("$name", d["value"], d["serial"]["ms_per_frame"], s["sort_total"], s["project"], s["binning"], s["composite"]))
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/${T}_$name.err").read()[-1500:])
PY
}
for rep in 1 2; do
one cfg3s -- --workload cfg3s
one cfg3s_nocompact MSPLAT_WS_COMPACT=0 -- --workload cfg3s
done
