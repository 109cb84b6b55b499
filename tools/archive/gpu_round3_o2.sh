#!/bin/bash
# passes 1 and 2 of the sort with 4096-key chunks when an earlier frame's V was small: parity + band table + scene-like workload
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03o2
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "large_cloud or sort or band or scene or cfg4 or in_flight" 2>&1 | tail -4
timeout 600 python tools/band_table.py --workload cfg4 --world 8 --layouts contiguous,block:8 --out gpurun_out/${T}_cfg4_bands.json 2>&1 | grep -v "^/opt" | tail -22
one() {  # name, env..., -- bench args
  name=$1; shift
  envs=(X=1)
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py --frames-in-flight 1 --no-cpu-baseline --steps 200 --warmup 50 --prewarm 100 --profile-frames 2 "$@" > gpurun_out/${T}_$name.json 2> gpurun_out/${T}_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${T}_$name.json").read().strip().splitlines()[-1])
    s = d["serial"]["stages_ms"]
    print("%-22s serial %.4f ms  sort %.4f  project %.4f  binning %.4f  comp %.4f (kernel %.4f)  latency %.4f" % ("$name", d["serial"]["ms_per_frame"], s["sort_total"], s["project"], s["binning"], s["composite"], s.get("composite_kernel", 0), d["serial"]["single_frame_latency_ms_host_to_host"]))
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/${T}_$name.err").read()[-1500:])
PY
}
one cfg3s -- --workload cfg3s --steps 60 --warmup 10 --prewarm 30
one cfg3 -- --workload cfg3 --steps 60 --warmup 10 --prewarm 30
