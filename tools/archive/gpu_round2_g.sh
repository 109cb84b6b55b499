#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
run() {
  tag=$1; shift
  timeout 600 python bench.py --steps 200 --warmup 20 --prewarm 50 --no-cpu-baseline "$@" > gpurun_out/r2g_$tag.json 2> gpurun_out/r2g_$tag.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r2g_$tag.json").read().strip().splitlines()[-1])
    s = d["serial"]["stages_ms"] or {}
    print("%-22s fps %7.1f | serial ms %.4f | sort %.4f proj %.4f bin %.4f comp %.4f" % ("$tag", d["value"], d["serial"]["ms_per_frame"], s.get("sort_total", 0), s.get("project", 0), s.get("binning", 0), s.get("composite", 0)))
except Exception as e:
    print("$tag failed:", e); print(open("gpurun_out/r2g_$tag.err").read()[-800:])
PY
}
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x 2>&1 | tail -3
for wl in cfg2 cfg4 cfg5; do run ${wl} --workload $wl; done
run cfg2_serial --frames-in-flight 1
