#!/bin/bash
# round 3, call D: after batching the loads of the sort / binning kernels and the cooperative pair emission
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03d
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rsP ) > gpurun_out/${T}_gpu_tests.log 2>&1
grep -aE "passed|failed|SKIPPED|^FAILED|^ERROR|scene-like" gpurun_out/${T}_gpu_tests.log | head -30
one() {  # name, env..., -- bench args
  name=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py --frames-in-flight 1 --no-cpu-baseline --steps 120 --warmup 30 --prewarm 100 --profile-frames 2 "$@" > gpurun_out/${T}_$name.json 2> gpurun_out/${T}_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${T}_$name.json").read().strip().splitlines()[-1])
    s = d["serial"]["stages_ms"]
    print("%-22s serial %.4f ms  sort %.4f  project %.4f  binning %.4f  comp %.4f (kernel %.4f)" % ("$name", d["serial"]["ms_per_frame"], s["sort_total"], s["project"], s["binning"], s["composite"], s.get("composite_kernel", 0)))
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/${T}_$name.err").read()[-1500:])
PY
}
prof() {  # name, bench args
  name=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_prof_$name -o run --output-format csv -- python $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/${T}_prof_$name.log 2>&1)
  f=$(find gpurun_out/${T}_prof_$name -name run_kernel_stats.csv | head -1)
  cp $f gpurun_out/${T}_${name}_kernel_stats.csv
  python - <<PY
import csv
print("== $name")
for r in csv.DictReader(open("gpurun_out/${T}_${name}_kernel_stats.csv")):
    if float(r["Percentage"]) > 0.4: print("   %-70s calls=%-5s avg=%8.1fus %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"][:5]))
PY
  rm -rf gpurun_out/${T}_prof_$name
}
L3="--workload cfg3 --steps 60 --warmup 10 --prewarm 30"
L4="--workload cfg4 --steps 40 --warmup 10 --prewarm 20"
L3S="--workload cfg3s --steps 60 --warmup 10 --prewarm 30"
one cfg2_base --
one cfg2_search MSPLAT_TILE_TABLE=search --
one cfg2_lsd8 MSPLAT_SORT=lsd8 --
one cfg2_ws16 MSPLAT_WS_ITEMS=16 --
one cfg3_base -- $L3
one cfg3_ws8 MSPLAT_WS_ITEMS=8 -- $L3
one cfg3_lsd8 MSPLAT_SORT=lsd8 -- $L3
one cfg4_base -- $L4
one cfg3s_base -- $L3S
prof cfg2_serial --frames-in-flight 1 --steps 300 --warmup 50 --prewarm 100
prof cfg3_serial --workload cfg3 --frames-in-flight 1 --steps 50 --warmup 10 --prewarm 30
prof cfg3s_serial --workload cfg3s --frames-in-flight 1 --steps 50 --warmup 10 --prewarm 30
prof cfg4_serial --workload cfg4 --frames-in-flight 1 --steps 30 --warmup 10 --prewarm 20
echo "== band table"
timeout 600 python tools/band_table.py --workload cfg4 --world 8 --layouts contiguous,block:4,block:8 --out gpurun_out/${T}_cfg4_bands.json 2>&1 | grep -v "    rank"
echo "== default bench"
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/${T}_cfg2_default.json 2> gpurun_out/${T}_cfg2_default.err
timeout 900 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/${T}_cfg2_steps20.json 2> gpurun_out/${T}_cfg2_steps20.err
python - <<PY
import json
for n in ("default", "steps20"):
    try:
        d = json.loads(open("gpurun_out/${T}_cfg2_%s.json" % n).read().strip().splitlines()[-1])
        print("%s: %.0f fps, serial %.0f fps (%.4f ms)" % (n, d["value"], d["serial"]["frames_per_sec"], d["serial"]["ms_per_frame"]))
    except Exception as e:
        print(n, "FAILED", e)
PY
