#!/bin/bash
# round 4, run h: SQ counters of the two compositor formulations (the matrix-pipe experiment), stereo in one chain
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
export TMPDIR=/tmp
T=r4h
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider -k "stereo or two_views or config5" 2>&1 | tail -8
for b in "" "--no-stereo-batch"; do
  for rep in 1 2; do
    timeout 600 python bench.py --workload cfg5 --steps 200 --warmup 20 --prewarm 100 --no-cpu-baseline $b > gpurun_out/${T}_cfg5_${rep}${b}.json 2> gpurun_out/${T}_cfg5_${rep}${b}.err
    python - <<PY
import json
d = json.load(open("gpurun_out/${T}_cfg5_${rep}${b}.json"))
print("cfg5 '$b' rep $rep: value %.0f  serial %.4f ms  stages %s  frac %.3f" % (d["value"], d["serial"]["ms_per_frame"], {k: round(v, 4) for k, v in d["serial"]["stages_ms"].items() if k != "frames_averaged"}, d["roofline"]["frac"]))
PY
  done
done
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u | head -20
OUT=gpurun_out/${T}_pmc_sq_compositor_mfma.txt
echo "# rocprofv3 --pmc, bench.py --frames-in-flight 1 (config 2), composite_kernel only; one counter group per pass; SQ_* cycle counters in quad-cycles" > $OUT
for c in wave mfma; do
  i=0
  for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    echo "## compositor = $c, group $i" >> $OUT
    MSPLAT_DEV_COMP=$c bash tools/gpu_pmc.sh ${T}_${c}_$i "$grp" --frames-in-flight 1 --prewarm 20 --serial-frames 8 2>&1 | grep -E "^kernel |composite_kernel" | cut -c1-300 >> $OUT
  done
done
cat $OUT
