#!/bin/bash
# one workload, variants from $AB_LIST ("tag ENV=..." per line): serial + default frames in flight
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
WL=${WL:-cfg4}
run() {
  tag=$1; shift
  env "$@" timeout 600 python bench.py --workload $WL --steps 100 --warmup 20 --prewarm 50 --no-cpu-baseline > gpurun_out/r2e_$tag.json 2> gpurun_out/r2e_$tag.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r2e_$tag.json").read().strip().splitlines()[-1])
    s = d["serial"]["stages_ms"] or {}
    print("%-16s $WL fps %7.1f | serial %.4f ms (sort %.4f proj %.4f bin %.4f comp %.4f compk %.4f) | overlapped compk %.4f" % (
        "$tag", d["value"], d["serial"]["ms_per_frame"], s.get("sort_total", 0), s.get("project", 0), s.get("binning", 0), s.get("composite", 0),
        s.get("composite_kernel", 0), d["stages_ms"].get("composite_kernel", 0)))
except Exception as e:
    print("$tag failed:", e); print(open("gpurun_out/r2e_$tag.err").read()[-800:])
PY
}
printf '%s\n' "${AB_LIST:-now}" | while read -r tag envs; do [ -n "$tag" ] && run $tag $envs; done
