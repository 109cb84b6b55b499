#!/bin/bash
# kernel durations of the one-partition binning (serial frames, config 2)
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03m2_prof -o run --output-format csv -- python $R/bench.py --no-cpu-baseline --frames-in-flight 1 --steps 100 --warmup 20 --prewarm 40 --serial-frames 16 --profile-frames 1 "$@" > $R/gpurun_out/r03m2_prof.log 2>&1)
f=$(find gpurun_out/r03m2_prof -name run_kernel_stats.csv | head -1)
python - <<PY
import csv
for r in csv.DictReader(open("$f")):
    if float(r["Percentage"]) > 0.5: print("   %-62s calls=%-5s avg=%8.1fus %5s%%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"][:5]))
PY
rm -rf gpurun_out/r03m2_prof
