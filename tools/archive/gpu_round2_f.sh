#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
run() {
  tag=$1; shift
  timeout 600 python bench.py --steps 100 --warmup 20 --prewarm 50 --no-cpu-baseline --frames-in-flight 1 "$@" > gpurun_out/r2f_$tag.json 2> gpurun_out/r2f_$tag.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r2f_$tag.json").read().strip().splitlines()[-1])
    s = d["serial"]["stages_ms"] or {}
    print("%-22s serial fps %7.1f ms %.4f | sort %.4f proj %.4f bin %.4f comp %.4f" % ("$tag", d["value"], d["ms_per_step"], s.get("sort_total", 0), s.get("project", 0), s.get("binning", 0), s.get("composite", 0)))
except Exception as e:
    print("$tag failed:", e); print(open("gpurun_out/r2f_$tag.err").read()[-800:])
PY
}
for wl in cfg3 cfg4; do
  for mc in 2048 4096 8192 16384; do
    MSPLAT_FUSED_MAX_CHUNKS=$mc run ${wl}_mc$mc --workload $wl
  done
done
