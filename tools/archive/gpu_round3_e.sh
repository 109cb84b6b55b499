#!/bin/bash
# round 3, call E: heavy / light pair emission, coalesced bin counts
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r03e
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rsP ) > gpurun_out/${T}_gpu_tests.log 2>&1
grep -aE "passed|failed|^FAILED|^ERROR" gpurun_out/${T}_gpu_tests.log | head -30
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
one cfg2_base --
one cfg2_search MSPLAT_TILE_TABLE=search --
one cfg3_base -- --workload cfg3 --steps 60 --warmup 10 --prewarm 30
one cfg4_base -- --workload cfg4 --steps 40 --warmup 10 --prewarm 20
one cfg4_search MSPLAT_TILE_TABLE=search -- --workload cfg4 --steps 40 --warmup 10 --prewarm 20
one cfg3s_base -- --workload cfg3s --steps 60 --warmup 10 --prewarm 30
one cfg5_base -- --workload cfg5 --steps 100 --warmup 20 --prewarm 50
prof cfg2_serial --frames-in-flight 1 --steps 300 --warmup 50 --prewarm 100
prof cfg3s_serial --workload cfg3s --frames-in-flight 1 --steps 50 --warmup 10 --prewarm 30
bash tools/gpu_pmc.sh ${T}_bin "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" --workload cfg3s --frames-in-flight 1 --prewarm 10 --serial-frames 6 2>&1 | grep -E "kernel |bin1|radix_up|radix_down|project" | cut -c1-250
bash tools/gpu_pmc.sh ${T}_bin2 "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" --workload cfg3s --frames-in-flight 1 --prewarm 10 --serial-frames 6 2>&1 | grep -E "kernel |bin1|radix_up|radix_down|project" | cut -c1-250
echo "== default bench"
timeout 900 python bench.py > gpurun_out/${T}_cfg2_default.json 2> gpurun_out/${T}_cfg2_default.err
python - <<PY
import json
d = json.loads(open("gpurun_out/${T}_cfg2_default.json").read().strip().splitlines()[-1])
print("default: %.0f fps, serial %.0f fps (%.4f ms)" % (d["value"], d["serial"]["frames_per_sec"], d["serial"]["ms_per_frame"]))
print("cpu_baseline", {k: d["cpu_baseline"][k] for k in ("value", "cores", "stages_ms", "thread_count_trials_sec")})
PY
