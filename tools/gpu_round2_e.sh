#!/bin/bash
# all BASELINE workloads: serial + default frames in flight, split-queue compositor on/off
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
run() {
  tag=$1; shift
  timeout 600 python bench.py --steps 200 --warmup 30 --no-cpu-baseline "$@" > gpurun_out/r2e_$tag.json 2> gpurun_out/r2e_$tag.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r2e_$tag.json").read().strip().splitlines()[-1])
    s = d["serial"]["stages_ms"] or {}
    r = d["roofline"]
    print("%-16s fps %7.1f | serial %.4f ms (sort %.4f proj %.4f bin %.4f comp %.4f compk %.4f) | roof frac %.3f formula %.3f valu %s" % (
        "$tag", d["value"], d["serial"]["ms_per_frame"], s.get("sort_total", 0), s.get("project", 0), s.get("binning", 0), s.get("composite", 0),
        s.get("composite_kernel", 0), r["frac"], r["formula_frac"] or 0, (r["valu"] or {}).get("frac_of_fp32_vector_peak")))
except Exception as e:
    print("$tag failed:", e); print(open("gpurun_out/r2e_$tag.err").read()[-800:])
PY
}
for wl in cfg2 cfg3 cfg4 cfg5; do
  for sq in 1 0; do
    MSPLAT_COMP_SPLITQ=$sq run ${wl}_sq$sq --workload $wl
  done
done
