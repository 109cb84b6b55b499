#!/bin/bash
# round 4, run d (live-box list, lambda_max bound, size classes): spatial storage order + chunk-level cull -- tests, a band rank's kernels, bench lines of the workloads it touches
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rP -x ) > gpurun_out/r4d_gpu_tests.log 2>&1
grep -E "passed|failed|error|SKIPPED" gpurun_out/r4d_gpu_tests.log | tail -5
grep -E "chunk-level cull|band cull at box level|scene-like 6M" gpurun_out/r4d_gpu_tests.log | head -12
grep -E "^(FAILED|ERROR)|^E  " gpurun_out/r4d_gpu_tests.log | head -30
prof() {  # name, command...
  name=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4d_prof_$name -o run --output-format csv -- "$@" > $R/gpurun_out/r4d_prof_$name.log 2>&1)
  f=$(find gpurun_out/r4d_prof_$name -name run_kernel_stats.csv | head -1)
  cp $f gpurun_out/r4d_${name}_kernel_stats.csv
  rm -rf gpurun_out/r4d_prof_$name
  python - <<PY
import csv
print("== $name")
for r in csv.DictReader(open("gpurun_out/r4d_${name}_kernel_stats.csv")):
    if float(r["Percentage"]) > 0.8: print("   %-70s calls=%-5s avg=%8.1fus %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"][:5]))
PY
  tail -2 gpurun_out/r4d_prof_$name.log
}
prof cfg4_rank3_of_8 python $R/tools/band_rank_profile.py cfg4 8 3 block 8 100
for wl in cfg2 cfg3s cfg4; do
  timeout 900 python bench.py --workload $wl --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r4d_${wl}_bench.json 2> gpurun_out/r4d_${wl}_bench.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r4d_${wl}_bench.json"))
print("$wl value %.0f serial %.4f ms stages %s" % (d["value"], d["serial"]["ms_per_frame"], {k: round(v, 4) for k, v in d["serial"]["stages_ms"].items() if k != "frames_averaged"}))
PY
done
