#!/bin/bash
# round 3, call H: two-level group tables (no radix_scan launches at 6 M), sort_mode per context
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03h
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rsP ) > gpurun_out/${T}_gpu_tests.log 2>&1
grep -aE "passed|failed|^FAILED|^ERROR" gpurun_out/${T}_gpu_tests.log | head -30
one() {  # name, env..., -- bench args
  name=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py --no-cpu-baseline --steps 120 --warmup 30 --prewarm 100 --profile-frames 2 "$@" > gpurun_out/${T}_$name.json 2> gpurun_out/${T}_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${T}_$name.json").read().strip().splitlines()[-1])
    s = d["serial"]["stages_ms"]
    print("%-22s %6.0f fps | serial %.4f ms  sort %.4f  project %.4f  binning %.4f  comp %.4f (kernel %.4f)" % ("$name", d["value"], d["serial"]["ms_per_frame"], s["sort_total"], s["project"], s["binning"], s["composite"], s.get("composite_kernel", 0)))
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/${T}_$name.err").read()[-1500:])
PY
}
L3="--workload cfg3 --steps 60 --warmup 10 --prewarm 30"
L4="--workload cfg4 --steps 40 --warmup 10 --prewarm 20"
L3S="--workload cfg3s --steps 60 --warmup 10 --prewarm 30"
one cfg2_base --
one cfg2_old MSPLAT_FUSED_MAX_CHUNKS=4096 --
one cfg3_base -- $L3
one cfg3_old MSPLAT_FUSED_MAX_CHUNKS=4096 -- $L3
one cfg4_base -- $L4
one cfg4_old MSPLAT_FUSED_MAX_CHUNKS=4096 -- $L4
one cfg3s_base -- $L3S
one cfg3s_old MSPLAT_FUSED_MAX_CHUNKS=4096 -- $L3S
one cfg5_base -- --workload cfg5 --steps 100 --warmup 20 --prewarm 50
