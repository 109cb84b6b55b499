#!/bin/bash
# GPU call B of round 2: correctness of the scan-free passes, the 64-ary tile_start and the four-wave compositor,
# then A/B timings of the variants (serial frames and 4 frames in flight), then a kernel trace of the serial default.
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
[ -n "$SKIP_TESTS" ] || ( time timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -rsP ) > gpurun_out/r2b_tests.log 2>&1
grep -E "passed|failed|FAILED|SKIPPED|check_image:|Error|error" gpurun_out/r2b_tests.log | head -60
ab() {   # tag, env...
  tag=$1; shift
  for P in 1 4; do
    env "$@" timeout 200 python bench.py --steps 300 --warmup 50 --no-cpu-baseline --frames-in-flight $P > gpurun_out/r2b_${tag}_p$P.json 2> gpurun_out/r2b_${tag}_p$P.err
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r2b_${tag}_p$P.json").read().strip().splitlines()[-1])
    s = d["serial"]["stages_ms"] or {}
    print("%-18s P=$P fps %7.1f  serial %.4f ms  lat %.4f | sort %.4f proj %.4f bin %.4f comp %.4f compk %.4f | roof frac %.3f valu %.3f" % (
        "$tag", d["value"], d["serial"]["ms_per_frame"], d["serial"]["single_frame_latency_ms_host_to_host"], s.get("sort_total", 0), s.get("project", 0),
        s.get("binning", 0), s.get("composite", 0), s.get("composite_kernel", 0), d["roofline"]["frac"], (d["roofline"]["valu"] or {}).get("frac_of_fp32_vector_peak", 0)))
except Exception as e:
    print("$tag P=$P failed:", e); print(open("gpurun_out/r2b_${tag}_p$P.err").read()[-1500:])
PY
  done
}
# variants to compare: lines of "tag ENV=..." in $AB_LIST (default: the build as it is, twice)
printf '%s\n' "${AB_LIST:-now
now_b}" | while read -r tag envs; do [ -n "$tag" ] && ab $tag $envs; done
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2b_prof -o run --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline --frames-in-flight 1 > $GRAFT_REPO_ROOT/gpurun_out/r2b_prof.log 2>&1)
python - <<PY
import csv, glob
for f in glob.glob("gpurun_out/r2b_prof/**/run_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.3: print("%-62s calls=%-5s avg=%8.1fus %5s%%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"][:5]))
PY
