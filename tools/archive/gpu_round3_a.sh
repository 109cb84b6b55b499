#!/bin/bash
# round 3, call A: the 3-pass wide-digit sort + bin counts in the row pass (no tile_start / tile_order launches).
# parity first, then A/B of the serial frame against the r2 paths (MSPLAT_SORT=lsd8, MSPLAT_TILE_TABLE=search).
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03a
( time timeout 300 python -m pytest tests/test_gpu_parity.py -q -x --timeout 300 -p no:cacheprovider -k "sort or tile_lists or image_matches or hard" ) > gpurun_out/${T}_quick.log 2>&1
tail -5 gpurun_out/${T}_quick.log
( time timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -rsP ) > gpurun_out/${T}_gpu_tests.log 2>&1
grep -E "passed|failed|SKIPPED|Error|error" gpurun_out/${T}_gpu_tests.log | head -20
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
one cfg2_lsd8 MSPLAT_SORT=lsd8 --
one cfg2_search MSPLAT_TILE_TABLE=search --
one cfg2_r2 MSPLAT_SORT=lsd8 MSPLAT_TILE_TABLE=search --
one cfg2_ws16 MSPLAT_WS_ITEMS=16 --
one cfg3_new -- --workload cfg3 --steps 60 --warmup 10 --prewarm 30
one cfg3_ws8 MSPLAT_WS_ITEMS=8 -- --workload cfg3 --steps 60 --warmup 10 --prewarm 30
one cfg3_r2 MSPLAT_SORT=lsd8 MSPLAT_TILE_TABLE=search -- --workload cfg3 --steps 60 --warmup 10 --prewarm 30
one cfg4_new -- --workload cfg4 --steps 40 --warmup 10 --prewarm 20
one cfg4_r2 MSPLAT_SORT=lsd8 MSPLAT_TILE_TABLE=search -- --workload cfg4 --steps 40 --warmup 10 --prewarm 20
prof() {  # name, bench args
  name=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_prof_$name -o run --output-format csv -- python $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/${T}_prof_$name.log 2>&1)
  f=$(find gpurun_out/${T}_prof_$name -name run_kernel_stats.csv | head -1)
  cp $f gpurun_out/${T}_${name}_kernel_stats.csv
  python - <<PY
import csv
print("== $name")
for r in csv.DictReader(open("gpurun_out/${T}_${name}_kernel_stats.csv")):
    if float(r["Percentage"]) > 0.5: print("   %-70s calls=%-5s avg=%8.1fus %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"][:5]))
PY
  rm -rf gpurun_out/${T}_prof_$name
}
prof cfg2_serial --frames-in-flight 1 --steps 400 --warmup 100 --prewarm 100
prof cfg3_serial --workload cfg3 --frames-in-flight 1 --steps 60 --warmup 10 --prewarm 30
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/${T}_cfg2_default.json 2> gpurun_out/${T}_cfg2_default.err
python -c "
import json; d=json.loads(open('gpurun_out/${T}_cfg2_default.json').read().strip().splitlines()[-1]); print('default bench: %.0f fps, serial %.0f fps' % (d['value'], d['serial']['frames_per_sec']))"
