#!/bin/bash
# r4 run l: row-pass downsweep with 512 / 1024 threads per chunk (4096- and 8192-word chunks): parity subset, serial binning stage
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
T=r4l
mkdir -p gpurun_out
for v in t512 t512c8k t1024c8k; do
  MSPLAT_LIB_PATH=$PWD/tools/bin/variants/libmsplat_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "lists or config2 or band or scan_free or image or ragged or empty" 2>&1 | tail -3
done
for wl in cfg4 cfg3s cfg2; do
for v in base t512 t512occ3 t512c8k t512c8kocc1 t1024c8k base; do
  MSPLAT_LIB_PATH=$PWD/tools/bin/variants/libmsplat_$v.so timeout 300 python bench.py --workload $wl --frames-in-flight 1 --steps 30 --warmup 5 --prewarm 20 --serial-frames 48 --no-cpu-baseline --profile-frames 1 > gpurun_out/${T}_${wl}_${v}.json 2> gpurun_out/${T}_${wl}_${v}.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${T}_${wl}_${v}.json"))
    s = d["serial"]["stages_ms"]
    print("$wl %-12s serial %.4f ms  binning %.4f  sort %.4f project %.4f composite %.4f" % ("$v", d["serial"]["ms_per_frame"], s["binning"], s["sort_total"], s["project"], s["composite"]))
except Exception as e:
    print("$wl $v failed", e)
PY
done
done
