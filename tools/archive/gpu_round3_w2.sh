#!/bin/bash
# per-kernel durations of one rank (3 of 8, blocks of 8 rows) of config 4's row-sharded frame
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03w2_prof -o run --output-format csv -- python $R/tools/band_rank_profile.py cfg4 8 3 block 8 120 > $R/gpurun_out/r03w2_prof.log 2>&1)
tail -2 gpurun_out/r03w2_prof.log
f=$(find gpurun_out/r03w2_prof -name run_kernel_stats.csv | head -1)
cp $f gpurun_out/r03_cfg4_rank3_of_8_kernel_stats.csv
python - <<PY
import csv
for r in csv.DictReader(open("$f")):
    if float(r["Percentage"]) > 0.5: print("   %-62s calls=%-5s avg=%8.1fus %5s%%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"][:5]))
PY
rm -rf gpurun_out/r03w2_prof
