#!/bin/bash
# round 5, run i: the compositor on a second stream per context again (tools/bin/variants/libmsplat_split.so, the experiment of run c
# rebuilt on the final tree), this time with enough hardware queues for 4 + 4 user streams next to torch's: was run c's collapse
# (3.8 k frames/s) the queue sharing, or the cross-stream events?
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {  # label, env..., -- args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 120 python bench.py --no-cpu-baseline --gpus 1 --steps 20 --warmup 5 --serial-frames 16 "$@" 2>gpurun_out/err_i.txt | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); s=d.get('serial',{})
    print('%-30s fps %.0f   serial %.4f ms  in-flight stages %s' % ('$label', d['value'], s.get('ms_per_frame', 0), {k: round(v, 3) for k, v in d.get('stages_ms', {}).items() if k != 'frames_averaged'}))
except Exception as e:
    print('$label', 'FAILED', e); print(open('gpurun_out/err_i.txt').read()[-600:])"
}
V=$PWD/tools/bin/variants/libmsplat_split.so
run base X=1 --
run base_q16 GPU_MAX_HW_QUEUES=16 --
run split_q16_low MSPLAT_LIB_PATH=$V MSPLAT_X_SPLIT=1 GPU_MAX_HW_QUEUES=16 --
run split_q16_low_mainhigh MSPLAT_LIB_PATH=$V MSPLAT_X_SPLIT=1 MSPLAT_X_MAIN_PRIO=-1 GPU_MAX_HW_QUEUES=16 --
run split_q16_norm MSPLAT_LIB_PATH=$V MSPLAT_X_SPLIT=1 MSPLAT_X_COMP_PRIO=0 GPU_MAX_HW_QUEUES=16 --
run split_q24_low_mainhigh MSPLAT_LIB_PATH=$V MSPLAT_X_SPLIT=1 MSPLAT_X_MAIN_PRIO=-1 GPU_MAX_HW_QUEUES=24 --
run split_q12_low_mainhigh MSPLAT_LIB_PATH=$V MSPLAT_X_SPLIT=1 MSPLAT_X_MAIN_PRIO=-1 GPU_MAX_HW_QUEUES=12 --
run split_q16_fif3 MSPLAT_LIB_PATH=$V MSPLAT_X_SPLIT=1 MSPLAT_X_MAIN_PRIO=-1 GPU_MAX_HW_QUEUES=16 -- --frames-in-flight 3
run split_q16_fif6 MSPLAT_LIB_PATH=$V MSPLAT_X_SPLIT=1 MSPLAT_X_MAIN_PRIO=-1 GPU_MAX_HW_QUEUES=24 -- --frames-in-flight 6
run base X=1 --
