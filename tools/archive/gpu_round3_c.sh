#!/bin/bash
# round 3, call C: full GPU suite; kernel profiles of every workload; A/B of the XCD-contiguous chunk ranges, the column
# pass's chunk size and the compositor's strip-pair mask; default bench with the auto-tuned CPU baseline.
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03c
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rsP ) > gpurun_out/${T}_gpu_tests.log 2>&1
grep -aE "passed|failed|SKIPPED|^FAILED|^ERROR|scene-like" gpurun_out/${T}_gpu_tests.log | head -30
one() {  # name, env..., -- bench args
  name=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py --frames-in-flight 1 --no-cpu-baseline --steps 120 --warmup 30 --prewarm 100 --profile-frames 2 "$@" > gpurun_out/${T}_$name.json 2> gpurun_out/${T}_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${T}_$name.json").read().strip().splitlines()[-1])
    s = d["serial"]["stages_ms"]
    print("%-22s serial %.4f ms  sort %.4f  project %.4f  binning %.4f  comp %.4f (kernel %.4f)" % ("$name", d["serial"]["ms_per_frame"], s["sort_total"], s["project"], s["binning"], s["composite"], s.get("composite_kernel", 0)))
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/${T}_$name.err").read()[-1500:])
PY
}
L3="--workload cfg3 --steps 60 --warmup 10 --prewarm 30"
L4="--workload cfg4 --steps 40 --warmup 10 --prewarm 20"
one cfg2_base --
one cfg2_smask MSPLAT_COMP_STRIPMASK=1 --
one cfg2_xcd1 MSPLAT_XCD_MAP=1 --
one cfg2_xcd2 MSPLAT_XCD_MAP=2 --
one cfg3_base -- $L3
one cfg3_xcd1 MSPLAT_XCD_MAP=1 -- $L3
one cfg3_xcd2 MSPLAT_XCD_MAP=2 -- $L3
one cfg3_bc1024 MSPLAT_BIN_CHUNK=1024 -- $L3
one cfg4_base -- $L4
one cfg4_xcd3 MSPLAT_XCD_MAP=3 -- $L4
one cfg4_bc1024 MSPLAT_BIN_CHUNK=1024 -- $L4
one cfg4_bc1024_xcd2 MSPLAT_BIN_CHUNK=1024 MSPLAT_XCD_MAP=2 -- $L4
one cfg4_smask MSPLAT_COMP_STRIPMASK=1 -- $L4
prof() {  # name, bench args
  name=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_prof_$name -o run --output-format csv -- python $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/${T}_prof_$name.log 2>&1)
  f=$(find gpurun_out/${T}_prof_$name -name run_kernel_stats.csv | head -1)
  cp $f gpurun_out/${T}_${name}_kernel_stats.csv
  python - <<PY
import csv
print("== $name")
for r in csv.DictReader(open("gpurun_out/${T}_${name}_kernel_stats.csv")):
    if float(r["Percentage"]) > 0.4: print("   %-70s calls=%-5s avg=%8.1fus %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"][:5]))
PY
  rm -rf gpurun_out/${T}_prof_$name
}
prof cfg2_serial --frames-in-flight 1 --steps 300 --warmup 50 --prewarm 100
prof cfg3_serial --workload cfg3 --frames-in-flight 1 --steps 50 --warmup 10 --prewarm 30
prof cfg4_serial --workload cfg4 --frames-in-flight 1 --steps 30 --warmup 10 --prewarm 20
prof cfg3s_serial --workload cfg3s --frames-in-flight 1 --steps 50 --warmup 10 --prewarm 30
echo "== default bench with the CPU baselines"
timeout 900 python bench.py > gpurun_out/${T}_cfg2_default.json 2> gpurun_out/${T}_cfg2_default.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${T}_cfg2_default.json").read().strip().splitlines()[-1])
    print("default: %.0f fps, serial %.0f fps" % (d["value"], d["serial"]["frames_per_sec"]))
    print("cpu_baseline", d.get("cpu_baseline")); print("cpu_baseline_literal", d.get("cpu_baseline_literal"))
except Exception as e:
    print("default FAILED", e); print(open("gpurun_out/${T}_cfg2_default.err").read()[-2500:])
PY
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os; print(len(os.sched_getaffinity(0)), os.cpu_count())"
