#!/bin/bash
# round 4, run e: listed pass 0 with 4096-key chunks; per-kernel stats of a band rank and of the scene-like workload; the
# 1/2/4/8-GPU prediction under bench.py's own protocol (4 frames in flight, 20-frame blocks)
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
export TMPDIR=/tmp
T=r4e
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x ) > gpurun_out/${T}_gpu_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/${T}_gpu_tests.log | tail -3
grep -E "^(FAILED|ERROR)|^E  " gpurun_out/${T}_gpu_tests.log | head -30
prof() {  # name, command...
  name=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_prof_$name -o run --output-format csv -- "$@" > $R/gpurun_out/${T}_prof_$name.log 2>&1)
  f=$(find gpurun_out/${T}_prof_$name -name run_kernel_stats.csv | head -1)
  cp $f gpurun_out/${T}_${name}_kernel_stats.csv
  rm -rf gpurun_out/${T}_prof_$name
  python - <<PY
import csv
print("== $name")
for r in csv.DictReader(open("gpurun_out/${T}_${name}_kernel_stats.csv")):
    if float(r["Percentage"]) > 0.8: print("   %-70s calls=%-5s avg=%8.1fus %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"][:5]))
PY
  grep -E "^V |^\{" gpurun_out/${T}_prof_$name.log | cut -c1-300 | tail -2
}
prof cfg4_rank3_of_8 python $R/tools/band_rank_profile.py cfg4 8 3 block 8 100
prof cfg3s_serial python $R/bench.py --workload cfg3s --frames-in-flight 1 --steps 100 --warmup 20 --prewarm 50 --no-cpu-baseline
timeout 900 python tools/band_table.py --workload cfg2 --fif 4 --block 20 --worlds 2,4,8 --layouts auto,contiguous --out gpurun_out/${T}_cfg2_bands_fif4.json > gpurun_out/${T}_cfg2_bands_fif4.log 2>&1
grep -E "^single|^G =" gpurun_out/${T}_cfg2_bands_fif4.log
timeout 1200 python tools/band_table.py --workload cfg4 --fif 4 --block 20 --worlds 2,4,8 --layouts auto,contiguous --out gpurun_out/${T}_cfg4_bands_fif4.json > gpurun_out/${T}_cfg4_bands_fif4.log 2>&1
grep -E "^single|^G =" gpurun_out/${T}_cfg4_bands_fif4.log
tail -3 gpurun_out/${T}_cfg4_bands_fif4.log
