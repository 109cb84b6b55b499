#!/bin/bash
# round 3, call I: wide stores in the column pass's emission (bin1_downsweep): binning parity tests + serial stage times
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03i
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "bin or tile or heavy or cfg4 or frame or band or overflow or capacity or scene" 2>&1 | tail -5
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
one cfg2 --
one cfg3 -- --workload cfg3 --steps 60 --warmup 10 --prewarm 30
one cfg4 -- --workload cfg4 --steps 40 --warmup 10 --prewarm 20
one cfg3s -- --workload cfg3s --steps 60 --warmup 10 --prewarm 30
one cfg5 -- --workload cfg5 --steps 100 --warmup 10 --prewarm 30
