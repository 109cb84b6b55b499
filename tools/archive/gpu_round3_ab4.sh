#!/bin/bash
# per-kernel durations with four frames in flight: round-2 tree vs head, same box
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
for t in r2 head; do
  if [ $t = r2 ]; then D=$R/build/r2tree; else D=$R; fi
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ab4_$t -o run --output-format csv -- python $D/bench.py --no-cpu-baseline --steps 600 --warmup 100 --serial-frames 8 --profile-frames 1 > $R/gpurun_out/ab4_$t.log 2>&1)
  f=$(find gpurun_out/ab4_$t -name run_kernel_stats.csv | head -1)
  cp $f gpurun_out/ab4_${t}_kernel_stats.csv
  rm -rf gpurun_out/ab4_$t
  python - <<PY
import csv, json
print("== $t", [json.loads(l)["value"] for l in open("gpurun_out/ab4_$t.log") if l.startswith("{")])
for r in csv.DictReader(open("gpurun_out/ab4_${t}_kernel_stats.csv")):
    if float(r["Percentage"]) > 0.6: print("   %-66s calls=%-5s avg=%8.1fus total=%7.1fms" % (r["Name"].replace("void msplat::","")[:66], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done
