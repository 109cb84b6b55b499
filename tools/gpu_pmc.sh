#!/bin/bash
# usage: tools_gpu_pmc.sh <tag> "<counters>" [bench args]   -- one rocprofv3 --pmc pass over a short bench run
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; CTRS=$2; shift; shift
cd /tmp && rocprofv3 --kernel-trace --pmc $CTRS -d $R/gpurun_out/pmc_$TAG -o run --output-format csv -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --profile-frames 1 --timing-stride 0 "$@" > $R/gpurun_out/pmc_$TAG.log 2>&1
ls $R/gpurun_out/pmc_$TAG | head
python - <<PY
import csv, collections
f="$R/gpurun_out/pmc_$TAG/run_counter_collection.csv"
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"][:48]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
names=sorted({c for k in agg for c in agg[k]})
disp=collections.Counter()
for r in csv.DictReader(open(f)):
    if r["Counter_Name"]==names[0]: disp[r["Kernel_Name"][:48]]+=1
print("kernel".ljust(50), " ".join(n[-18:].rjust(18) for n in names), "  (per dispatch)")
for k in sorted(agg, key=lambda k:-sum(agg[k].values())):
    d=max(disp[k],1)
    print(k.ljust(50), " ".join(("%.4g"%(agg[k][n]/d)).rjust(18) for n in names))
PY
