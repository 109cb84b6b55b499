#!/bin/bash
# r4 run k: row-pass variants (chunk size, occupancy hint, nontemporal loads / stores), serial binning stage at 6 M / 4096^2
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
T=r4k
mkdir -p gpurun_out
for v in base items32 items8 occ3 occ5 nt1 nt2 nt3 base; do
  MSPLAT_LIB_PATH=$PWD/tools/bin/variants/libmsplat_$v.so timeout 300 python bench.py --workload cfg4 --frames-in-flight 1 --steps 30 --warmup 5 --prewarm 20 --serial-frames 48 --no-cpu-baseline --profile-frames 1 > gpurun_out/${T}_cfg4_${v}.json 2> gpurun_out/${T}_cfg4_${v}.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${T}_cfg4_${v}.json"))
    s = d["serial"]["stages_ms"]
    print("cfg4 %-8s serial %.4f ms  binning %.4f  sort %.4f project %.4f composite %.4f" % ("$v", d["serial"]["ms_per_frame"], s["binning"], s["sort_total"], s["project"], s["composite"]))
except Exception as e:
    print("cfg4 $v failed", e)
PY
done
