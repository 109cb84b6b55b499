#!/bin/bash
# round 3, call J: XCD-contiguous chunk ranges for the row pass of the binning (MSPLAT_XCD_MAP bit 2), serial stage times
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03j
MSPLAT_XCD_MAP=5 timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "bin or tile or heavy or cfg4 or band" 2>&1 | tail -3
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
for rep in 1 2; do
one cfg4 -- --workload cfg4 --steps 40 --warmup 10 --prewarm 20
one cfg4_x5 MSPLAT_XCD_MAP=5 -- --workload cfg4 --steps 40 --warmup 10 --prewarm 20
one cfg4_x7 MSPLAT_XCD_MAP=7 -- --workload cfg4 --steps 40 --warmup 10 --prewarm 20
done
one cfg2 --
one cfg2_x5 MSPLAT_XCD_MAP=5 --
one cfg3 -- --workload cfg3 --steps 60 --warmup 10 --prewarm 30
one cfg3_x5 MSPLAT_XCD_MAP=5 -- --workload cfg3 --steps 60 --warmup 10 --prewarm 30
one cfg3s -- --workload cfg3s --steps 60 --warmup 10 --prewarm 30
one cfg3s_x5 MSPLAT_XCD_MAP=5 -- --workload cfg3s --steps 60 --warmup 10 --prewarm 30
