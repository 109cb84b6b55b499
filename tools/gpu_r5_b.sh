#!/bin/bash
# round 5, run b: (run a lost 25 GPU-minutes to a rocprofv3 that hung on the aborting r5 library: every command has a timeout now)
# sanity, GPU suite, same-box A/B of r3's tree (build/r3tree = 5e2f411) against this one, serial kernel stats
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 120 python bench.py --workload tiny --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300 || exit 1
timeout 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -x 2>&1 | tail -8
run() {  # label dir args...
  label=$1; dir=$2; shift; shift
  (cd $dir && timeout 120 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d.get('serial',{})
print('%-14s fps %.0f   serial %.4f ms  stages %s  compk %s' % ('$label', d['value'], s.get('ms_per_frame', 0), {k: round(v, 4) for k, v in s.get('stages_ms', {}).items()}, d['roofline'].get('avg_launch_ms')))")
}
for rep in 1 2 3; do
  run r3_steps20 build/r3tree --gpus 1 --steps 20 --warmup 5
  run r5_steps20 . --gpus 1 --steps 20 --warmup 5
done
for rep in 1 2; do
  run r3_500 build/r3tree --steps 500 --warmup 50
  run r5_500 . --steps 500 --warmup 50
done
prof() {  # tag dir bench-args
  tag=$1; d=$2; shift; shift
  (cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o run --output-format csv -- python $d/bench.py --steps 200 --warmup 50 --no-cpu-baseline --profile-frames 1 "$@" > $R/gpurun_out/prof_$tag.log 2>&1)
  python - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/prof_$tag/**/run_kernel_stats.csv", recursive=True)
print("== $tag kernel stats")
for r in csv.DictReader(open(f[0])):
    if float(r["Percentage"]) > 1.0: print("%-70s calls=%-5s avg=%8.1fus %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"][:5]))
PY
}
prof r5_serial $R --frames-in-flight 1
prof r5_fif4 $R
prof r3_fif4 $R/build/r3tree
