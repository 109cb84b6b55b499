#!/bin/bash
# sort upsweeps with twice the threads per chunk: parity, then A/B (serial stage times, in-flight frames/s)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03q2
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "large_cloud or sort or in_flight or frame_modes" 2>&1 | tail -3
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
run() {  # label, env, args
  label=$1; envs=$2; shift; shift
  env $envs timeout 300 python bench.py --no-cpu-baseline --serial-frames 8 --profile-frames 1 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-22s fps %.0f' % ('$label', d['value']))"
}
for rep in 1 2; do
one cfg2_up2 --
one cfg2_up1 MSPLAT_WS_UP2=0 --
one cfg3_up2 -- --workload cfg3 --steps 60 --warmup 10 --prewarm 30
one cfg3_up1 MSPLAT_WS_UP2=0 -- --workload cfg3 --steps 60 --warmup 10 --prewarm 30
run fif_up2_steps20 X=1 --steps 20 --warmup 5
run fif_up1_steps20 MSPLAT_WS_UP2=0 --steps 20 --warmup 5
run fif_up2_500 X=1 --steps 500 --warmup 50
run fif_up1_500 MSPLAT_WS_UP2=0 --steps 500 --warmup 50
done
