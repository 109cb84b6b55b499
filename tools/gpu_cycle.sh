#!/bin/bash
# usage (on the GPU box, from the repo root): tools_gpu_cycle.sh <tag> [bench args...]
# runs the GPU parity tests, then a rocprofv3 kernel-trace of bench.py, prints the JSON line + per-kernel stats
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
[ -n "$SKIP_TESTS" ] || timeout 600 python -m pytest $R/tests -m gpu -q --timeout 300 -p no:cacheprovider -x 2>&1 | tail -12
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o run --output-format csv -- python $R/bench.py --steps 600 --warmup 200 --no-cpu-baseline --profile-frames 4 "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
grep -h "^{" $R/gpurun_out/prof_$TAG.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('fps %.1f  ms %.4f  stages %s  roofline frac %.3f  V %.0f D %.0f' % (d['value'], d['ms_per_step'], {k: round(v,4) for k,v in d['stages_ms'].items()}, d['roofline']['frac'], d['config']['visible_V'], d['config']['pairs_D']))
"
python - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/prof_$TAG/run_kernel_stats.csv")):
    if float(r["Percentage"]) > 0.3: print("%-62s calls=%-4s avg=%8.1fus %5s%%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"][:5]))
PY
