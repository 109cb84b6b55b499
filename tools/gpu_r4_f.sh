#!/bin/bash
# round 4, run f: async_submit A/B under the driver's protocol; XCD-grouped chunk mapping of the column pass (item 3)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
T=r4f
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x ) > gpurun_out/${T}_gpu_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/${T}_gpu_tests.log | tail -3
grep -E "^(FAILED|ERROR)|^E  " gpurun_out/${T}_gpu_tests.log | head -30
for rep in 1 2; do
  for a in 0 1; do
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --async-submit $a > gpurun_out/${T}_async${a}_${rep}.json 2> gpurun_out/${T}_async${a}_${rep}.err
    python - <<PY
import json
d = json.load(open("gpurun_out/${T}_async${a}_${rep}.json"))
print("async $a rep $rep: value %.0f  host handover %.4f ms/frame  blocks %d  serial %.4f" % (d["value"], d["host_enqueue_ms_per_frame"], d["timed_blocks"], d["serial"]["ms_per_frame"]))
PY
  done
done
timeout 600 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline > gpurun_out/${T}_async1_long.json 2>/dev/null
python -c "import json; d=json.load(open('gpurun_out/${T}_async1_long.json')); print('async 1, 1000-frame blocks: %.0f' % d['value'])"
for wl in cfg4 cfg3s cfg2; do
  for g in 0 2 4 8; do
    MSPLAT_DEV_XCDG=$g timeout 600 python bench.py --workload $wl --frames-in-flight 1 --steps 60 --warmup 10 --prewarm 30 --serial-frames 64 --no-cpu-baseline --profile-frames 1 > gpurun_out/${T}_xcdg_${wl}_${g}.json 2>/dev/null
    python - <<PY
import json
d = json.load(open("gpurun_out/${T}_xcdg_${wl}_${g}.json"))
print("$wl xcdg $g: serial %.4f ms  binning %.4f" % (d["serial"]["ms_per_frame"], d["serial"]["stages_ms"]["binning"]))
PY
  done
done
