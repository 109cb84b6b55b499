#!/bin/bash
# round 4, run g: compositor with the exponents from the matrix pipe (MSPLAT_DEV_COMP=mfma): layout probe, parity subset, A/B
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
T=r4g
mkdir -p gpurun_out
tools/bin/ubench_mfma4x4 | tail -9
MSPLAT_DEV_COMP=mfma timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider -k "image or compositor or config2_sort or config5 or fp16 or frames_in_flight_bit or seeded_random" 2>&1 | tail -12
for rep in 1 2; do
  for c in wave mfma; do
    MSPLAT_DEV_COMP=$c timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/${T}_${c}_${rep}.json 2> gpurun_out/${T}_${c}_${rep}.err
    python - <<PY
import json
d = json.load(open("gpurun_out/${T}_${c}_${rep}.json"))
print("$c rep $rep: value %.0f  serial %.4f ms  composite_kernel serial %.4f  overlapped %.4f" % (d["value"], d["serial"]["ms_per_frame"], d["serial"]["stages_ms"]["composite_kernel"], d["stages_ms"]["composite_kernel"]))
PY
  done
done
for wl in cfg4 cfg3s cfg5; do
  for c in wave mfma; do
    MSPLAT_DEV_COMP=$c timeout 600 python bench.py --workload $wl --steps 100 --warmup 10 --prewarm 50 --serial-frames 64 --no-cpu-baseline --profile-frames 1 > gpurun_out/${T}_${wl}_${c}.json 2>/dev/null
    python - <<PY
import json
d = json.load(open("gpurun_out/${T}_${wl}_${c}.json"))
print("$wl $c: value %.0f  serial %.4f ms  composite_kernel serial %.4f" % (d["value"], d["serial"]["ms_per_frame"], d["serial"]["stages_ms"]["composite_kernel"]))
PY
  done
done
for g in 8 16 32; do
  MSPLAT_DEV_XCDG=$g timeout 600 python bench.py --workload cfg4 --frames-in-flight 1 --steps 60 --warmup 10 --prewarm 30 --serial-frames 64 --no-cpu-baseline --profile-frames 1 > gpurun_out/${T}_xcdg_cfg4_${g}.json 2>/dev/null
  python -c "import json; d=json.load(open('gpurun_out/${T}_xcdg_cfg4_${g}.json')); print('cfg4 xcdg $g: binning %.4f' % d['serial']['stages_ms']['binning'])"
done
