#!/bin/bash
# round 3: one-partition binning (bw_*): parity tests first, then serial stage times against the two-pass path
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03l2
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "one_partition or wide_sort_and_legacy" 2>&1 | tail -15
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -8
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
one cfg2_onepass --
one cfg2_twopass MSPLAT_BINNING=twopass --
one cfg5_onepass -- --workload cfg5 --steps 100 --warmup 10 --prewarm 30
one cfg5_twopass MSPLAT_BINNING=twopass -- --workload cfg5 --steps 100 --warmup 10 --prewarm 30
one cfg3_onepass MSPLAT_BINNING=onepass -- --workload cfg3 --steps 60 --warmup 10 --prewarm 30
one cfg3_twopass -- --workload cfg3 --steps 60 --warmup 10 --prewarm 30
