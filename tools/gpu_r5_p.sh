#!/bin/bash
# round 5, run p: the compositors of the frames in flight ONE AFTER THE OTHER (each waits for the event recorded behind the previous
# frame's compositor, tools/archive/r5_compositor_gate.patch), 4 / 5 / 6 frames in flight: does a compositor that always has the
# VALUs to itself, with the other frames' chains underneath, beat four frames that collide?
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {  # label, env..., -- args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 200 python bench.py --no-cpu-baseline --gpus 1 --serial-frames 16 "$@" 2>gpurun_out/err_p.txt | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read())
    print('%-18s fps %.0f   in flight: project %.3f composite %.3f composite_kernel %.3f sort %.3f binning %.3f' % ('$label', d['value'], d['stages_ms']['project'], d['stages_ms']['composite'], d['stages_ms']['composite_kernel'], d['stages_ms']['sort_total'], d['stages_ms']['binning']))
except Exception as e:
    print('$label', 'FAILED', e); print(open('gpurun_out/err_p.txt').read()[-600:])"
}
V=$PWD/tools/bin/variants/libmsplat_gate.so
run base X=1 -- --steps 20 --warmup 5
run gate_fif4 MSPLAT_LIB_PATH=$V MSPLAT_X_COMP_GATE=1 -- --steps 20 --warmup 5
run gate_fif5 MSPLAT_LIB_PATH=$V MSPLAT_X_COMP_GATE=1 GPU_MAX_HW_QUEUES=12 -- --steps 20 --warmup 5 --frames-in-flight 5
run gate_fif6 MSPLAT_LIB_PATH=$V MSPLAT_X_COMP_GATE=1 GPU_MAX_HW_QUEUES=12 -- --steps 20 --warmup 5 --frames-in-flight 6
run gate_fif4_100 MSPLAT_LIB_PATH=$V MSPLAT_X_COMP_GATE=1 -- --steps 100 --warmup 20
run gate_fif6_120 MSPLAT_LIB_PATH=$V MSPLAT_X_COMP_GATE=1 GPU_MAX_HW_QUEUES=12 -- --steps 120 --warmup 24 --frames-in-flight 6
run base_100 X=1 -- --steps 100 --warmup 20
