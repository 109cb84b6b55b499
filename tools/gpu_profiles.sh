#!/bin/bash
# usage (GPU box, repo root): tools/gpu_profiles.sh <round tag, e.g. r03>
# The evidence set of a round: GPU tests, the bench lines (default = 4 frames in flight, and serial), rocprofv3 kernel
# summaries of the same commands for every BASELINE workload, and the PMC traffic of the serial run.  Everything lands
# in gpurun_out/<tag>_*; copy what is to be judged into profiles/.
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
export TMPDIR=/tmp
T=${1:-r05}
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -rsP ) > gpurun_out/${T}_gpu_tests.log 2>&1
grep -E "passed|failed|SKIPPED|check_image:" gpurun_out/${T}_gpu_tests.log | head -20
( time timeout 600 python tools/soak.py --frames 3000 ) > gpurun_out/${T}_soak.log 2>&1
tail -4 gpurun_out/${T}_soak.log
timeout 600 python bench.py > gpurun_out/${T}_cfg2_bench.json 2> gpurun_out/${T}_cfg2_bench.err
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_cfg2_bench_steps20.json 2> gpurun_out/${T}_cfg2_bench_steps20.err
timeout 600 python bench.py --frames-in-flight 1 --no-cpu-baseline > gpurun_out/${T}_cfg2_bench_serial.json 2> gpurun_out/${T}_cfg2_bench_serial.err
prof() {  # name, bench args
  name=$1; shift
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_prof_$name -o run --output-format csv -- python $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/${T}_prof_$name.log 2>&1)   # (a rocprofv3 whose child aborts can hang for ever: r5 lost 25 GPU-minutes to one)
  f=$(find gpurun_out/${T}_prof_$name -name run_kernel_stats.csv | head -1)
  cp $f gpurun_out/${T}_${name}_kernel_stats.csv
  grep -h "^{" gpurun_out/${T}_prof_$name.log > gpurun_out/${T}_${name}_bench_under_rocprof.json
  rm -rf gpurun_out/${T}_prof_$name          # the per-dispatch trace: only the summary is kept
  python - <<PY
import csv
print("== $name")
for r in csv.DictReader(open("gpurun_out/${T}_${name}_kernel_stats.csv")):
    if float(r["Percentage"]) > 0.8: print("   %-62s calls=%-5s avg=%8.1fus %5s%%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"][:5]))
PY
}
prof cfg2_serial --frames-in-flight 1 --steps 600 --warmup 100
prof cfg2_fif4 --steps 600 --warmup 100
for wl in cfg3 cfg4 cfg5 cfg3s; do
  prof ${wl}_serial --workload $wl --frames-in-flight 1 --steps 100 --warmup 20 --prewarm 50
  timeout 600 python bench.py --workload $wl --no-cpu-baseline --steps 200 --warmup 30 > gpurun_out/${T}_${wl}_bench.json 2> gpurun_out/${T}_${wl}_bench.err
done
timeout 600 python tools/band_table.py --workload cfg4 --world 8 --layouts contiguous,block:4,block:8 --out gpurun_out/${T}_cfg4_bands.json > gpurun_out/${T}_cfg4_bands.log 2>&1
grep -v "    rank" gpurun_out/${T}_cfg4_bands.log | tail -6
# the 1/2/4/8-GPU prediction under bench.py's own protocol (4 frames in flight, blocks of 20 frames)
for wl in cfg2 cfg4; do
  timeout 1200 python tools/band_table.py --workload $wl --fif 4 --block 20 --worlds 2,4,8 --layouts auto,contiguous,weighted --out gpurun_out/${T}_${wl}_bands_fif4.json > gpurun_out/${T}_${wl}_bands_fif4.log 2>&1
  grep -E "^single|^G =" gpurun_out/${T}_${wl}_bands_fif4.log
done
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_prof_rank3 -o run --output-format csv -- python $R/tools/band_rank_profile.py cfg4 8 3 block 8 100 > $R/gpurun_out/${T}_prof_rank3.log 2>&1)
cp $(find gpurun_out/${T}_prof_rank3 -name run_kernel_stats.csv | head -1) gpurun_out/${T}_cfg4_rank3_of_8_kernel_stats.csv; rm -rf gpurun_out/${T}_prof_rank3
for wl in cfg2 cfg4; do
  bash tools/pmc_traffic.sh ${T}_$wl $wl
  cp gpurun_out/pmc_${T}_$wl/traffic.json gpurun_out/${T}_pmc_traffic_$wl.json
done
for C in FETCH_SIZE WRITE_SIZE; do cp $(find gpurun_out/pmc_${T}_cfg2/$C -name run_counter_collection.csv | head -1) gpurun_out/${T}_pmc_${C}_cfg2.csv; done
rm -rf gpurun_out/pmc_${T}_cfg2 gpurun_out/pmc_${T}_cfg4
