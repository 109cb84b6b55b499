#!/bin/bash
# three-pass sort with 4 keys per thread (twice the waves at 1 M splats): parity, then serial and in-flight A/B
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03n2
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "wide_sort_and_legacy" 2>&1 | tail -3
MSPLAT_WS_ITEMS=4 timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "sort_exact or in_flight or frame_modes" 2>&1 | tail -3
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
one cfg2_items8 --
one cfg2_items4 MSPLAT_WS_ITEMS=4 --
one cfg2_items4_t256 MSPLAT_WS_ITEMS=4 MSPLAT_WS_THREADS=256 --
run fif_items8_steps20 X=1 --steps 20 --warmup 5
run fif_items4_steps20 MSPLAT_WS_ITEMS=4 --steps 20 --warmup 5
run fif_items8_500 X=1 --steps 500 --warmup 50
run fif_items4_500 MSPLAT_WS_ITEMS=4 --steps 500 --warmup 50
done
