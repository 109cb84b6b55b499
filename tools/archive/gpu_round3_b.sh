#!/bin/bash
# round 3, call B: full GPU suite with the new tests (band layouts, device group, key ranges, whole-frame poses, scene-like 6M),
# binning A/B after the one-column fast path, the band table (predicted 8-GPU frame per layout), priority modes, cfg3s, CPU baseline.
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03b
( time timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rsP -x ) > gpurun_out/${T}_gpu_tests.log 2>&1
grep -aE "passed|failed|SKIPPED|Error|error|scene-like|check_image" gpurun_out/${T}_gpu_tests.log | head -30
tail -5 gpurun_out/${T}_gpu_tests.log
one() {  # name, env..., -- bench args
  name=$1; shift
  envs=()
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
one cfg2_new --
one cfg2_search MSPLAT_TILE_TABLE=search --
one cfg3_new -- --workload cfg3 --steps 60 --warmup 10 --prewarm 30
one cfg3_search MSPLAT_TILE_TABLE=search -- --workload cfg3 --steps 60 --warmup 10 --prewarm 30
one cfg4_new -- --workload cfg4 --steps 40 --warmup 10 --prewarm 20
one cfg4_search MSPLAT_TILE_TABLE=search -- --workload cfg4 --steps 40 --warmup 10 --prewarm 20
echo "== band tables"
timeout 600 python tools/band_table.py --workload cfg4 --world 8 --out gpurun_out/${T}_cfg4_bands.json 2>&1 | tail -60
timeout 300 python tools/band_table.py --workload cfg2 --world 8 --layouts contiguous,interleaved,block:2 --out gpurun_out/${T}_cfg2_bands.json 2>&1 | grep -v "    rank"
echo "== priority modes, 4 frames in flight"
for P in 0 1 2 0 1 2; do
  MSPLAT_COMP_PRIO=$P timeout 300 python bench.py --steps 500 --warmup 50 --no-cpu-baseline --serial-frames 8 --profile-frames 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prio $P  fps %.0f' % d['value'])"
done
echo "== cfg3s"
timeout 900 python bench.py --workload cfg3s --steps 60 --warmup 10 --prewarm 30 --no-cpu-baseline > gpurun_out/${T}_cfg3s.json 2> gpurun_out/${T}_cfg3s.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${T}_cfg3s.json").read().strip().splitlines()[-1])
    print("cfg3s: %.0f fps (4 in flight), serial %.3f ms" % (d["value"], d["serial"]["ms_per_frame"]), d["serial"]["stages_ms"])
    print({k: d["config"][k] for k in ("visible_V", "pairs_D", "pairs_binned_32px", "drawn", "D_over_N", "longest_bin_list", "pair_capacity", "pair_capacity_initial", "cameras")})
except Exception as e:
    print("cfg3s FAILED", e); print(open("gpurun_out/${T}_cfg3s.err").read()[-2500:])
PY
echo "== default bench with the CPU baselines"
timeout 900 python bench.py > gpurun_out/${T}_cfg2_default.json 2> gpurun_out/${T}_cfg2_default.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${T}_cfg2_default.json").read().strip().splitlines()[-1])
    print("default: %.0f fps, serial %.0f fps" % (d["value"], d["serial"]["frames_per_sec"]))
    print("cpu_baseline", d.get("cpu_baseline")); print("cpu_baseline_literal", d.get("cpu_baseline_literal"))
except Exception as e:
    print("default FAILED", e); print(open("gpurun_out/${T}_cfg2_default.err").read()[-2500:])
PY
