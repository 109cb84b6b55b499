#!/bin/bash
# usage (GPU box, repo root): tools/pmc_traffic.sh <tag> [workload]
# Collects HBM traffic of every kernel with rocprofv3 PMC counters, one counter family per pass
# (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2 -- MI355X_MICROARCH.md), no trace domains mixed in,
# over `bench.py --frames-in-flight 1` (every kernel alone on the GPU), and writes
# gpurun_out/pmc_<tag>/traffic.json (+ the raw counter CSVs).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1
WL=${2:-cfg2}
mkdir -p $R/gpurun_out/pmc_$TAG
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_$TAG/$C -o run --output-format csv -- \
     python $R/bench.py --workload $WL --frames-in-flight 1 --steps 8 --warmup 2 --prewarm 8 --serial-frames 8 --no-cpu-baseline --profile-frames 1 --timing-stride 0 > $R/gpurun_out/pmc_$TAG/$C.log 2>&1)
done
python - <<PY
import csv, json, collections, glob
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(float); n = collections.Counter()
    f = glob.glob("$R/gpurun_out/pmc_$TAG/%s/**/run_counter_collection.csv" % c, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k] += float(r["Counter_Value"]); n[k] += 1
    for k in agg:
        out.setdefault(k, {})[c + "_KB_per_launch"] = agg[k] / n[k]
        out[k]["launches"] = n[k]
# gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 128-B requests as 64 B for wide
# (16 B/lane) loads -> x2.  Calibrated on project_kernel, whose gather volume is known exactly.
for k, v in out.items():
    f = v.get("FETCH_SIZE_KB_per_launch", 0.0); w = v.get("WRITE_SIZE_KB_per_launch", 0.0)
    v["hbm_bytes_per_launch_corrected"] = (2.0 * f + w) * 1024.0
json.dump(out, open("$R/gpurun_out/pmc_$TAG/traffic.json", "w"), indent=1, sort_keys=True)
for k in sorted(out, key=lambda k: -out[k]["hbm_bytes_per_launch_corrected"])[:14]:
    print("%-52s fetch %9.0f KB  write %9.0f KB  corrected %7.1f MB  (%d launches)" % (k[:52], out[k].get("FETCH_SIZE_KB_per_launch", 0), out[k].get("WRITE_SIZE_KB_per_launch", 0), out[k]["hbm_bytes_per_launch_corrected"] / 1e6, out[k]["launches"]))
PY
