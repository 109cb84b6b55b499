#!/bin/bash
# round 5, run c: do frames in flight co-schedule better with (a) fewer persistent compositor waves,
# (b) a compositor whose waves per CU are capped by LDS, (c) the compositor on a low-priority stream of its own,
# (d) s_setprio 3 in the short sort / binning kernels, (e) other depths / submission modes?  Driver protocol (20-frame blocks).
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {  # label, env..., -- args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 120 python bench.py --no-cpu-baseline --gpus 1 --steps 20 --warmup 5 --serial-frames 16 "$@" 2>gpurun_out/err_c.txt | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); s=d.get('serial',{})
    print('%-26s fps %.0f   serial %.4f ms  in-flight stages %s' % ('$label', d['value'], s.get('ms_per_frame', 0), {k: round(v, 3) for k, v in d.get('stages_ms', {}).items() if k != 'frames_averaged'}))
except Exception as e:
    print('$label', 'FAILED', e); print(open('gpurun_out/err_c.txt').read()[-600:])"
}
V=$PWD/tools/bin/variants
run base X=1 --
run base X=1 --
MSPLAT_X_VERBOSE=1 MSPLAT_X_SPLIT=1 timeout 60 python bench.py --workload tiny --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep "priority range" | head -1
for w in 2048 3072 4096 6144; do run comp_waves_$w MSPLAT_COMP_WAVES=$w --; done
for l in 5000 7000 10500; do run comp_lds_$l MSPLAT_X_COMP_LDS=$l --; done
run split_low MSPLAT_X_SPLIT=1 --
run split_low_mainhigh MSPLAT_X_SPLIT=1 MSPLAT_X_MAIN_PRIO=-1 --
run split_norm_mainhigh MSPLAT_X_SPLIT=1 MSPLAT_X_COMP_PRIO=0 MSPLAT_X_MAIN_PRIO=-1 --
run split_norm MSPLAT_X_SPLIT=1 MSPLAT_X_COMP_PRIO=0 --
run prio3 MSPLAT_LIB_PATH=$V/libmsplat_prio3.so --
run prio3_split_low MSPLAT_LIB_PATH=$V/libmsplat_prio3.so MSPLAT_X_SPLIT=1 MSPLAT_X_MAIN_PRIO=-1 --
run base X=1 --
for p in 2 3 5 6 8; do run fif_$p X=1 -- --frames-in-flight $p; done
run fif_6_split MSPLAT_X_SPLIT=1 MSPLAT_X_MAIN_PRIO=-1 -- --frames-in-flight 6
run async0 X=1 -- --async-submit 0
run twopass_off X=1 -- --two-pass off
run base X=1 --
