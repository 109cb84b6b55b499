#!/bin/bash
# r4 run p: per-kernel durations of two-pass frames (rocprofv3 --kernel-trace --stats), one frame at a time
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
WL=${1:-cfg2}
mkdir -p $R/gpurun_out
rm -rf $R/gpurun_out/r4p_$WL
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4p_$WL -o run --output-format csv -- python $R/bench.py --workload $WL --two-pass on --frames-in-flight 1 --steps 100 --warmup 30 --prewarm 60 --serial-frames 64 --no-cpu-baseline --profile-frames 1 > $R/gpurun_out/r4p_$WL.log 2>&1)
f=$(find $R/gpurun_out/r4p_$WL -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:24]:
    print(r["Name"][:84].ljust(84), r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
PY
