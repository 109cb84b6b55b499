#!/bin/bash
# round 5, run g: with the priorities in, are r3's choices for in-flight contexts still right?  (list offsets from the search kernels
# vs the row pass's counts; 8-bit passes; more frames in flight with more hardware queues; comp waves)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {  # label, env..., -- args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 120 python bench.py --no-cpu-baseline --gpus 1 --steps 20 --warmup 5 --serial-frames 16 "$@" 2>gpurun_out/err_g.txt | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); s=d.get('serial',{})
    print('%-26s fps %.0f   serial %.4f ms  in-flight stages %s' % ('$label', d['value'], s.get('ms_per_frame', 0), {k: round(v, 3) for k, v in d.get('stages_ms', {}).items() if k != 'frames_averaged'}))
except Exception as e:
    print('$label', 'FAILED', e); print(open('gpurun_out/err_g.txt').read()[-600:])"
}
run base X=1 --
run base X=1 --
run tile_count MSPLAT_TILE_TABLE=count --
run tile_count MSPLAT_TILE_TABLE=count --
run sort_lsd8 MSPLAT_SORT=lsd8 --
run comp_4096 MSPLAT_COMP_WAVES=4096 --
run comp_2048 MSPLAT_COMP_WAVES=2048 --
run fif5_q12 GPU_MAX_HW_QUEUES=12 -- --frames-in-flight 5
run fif6_q12 GPU_MAX_HW_QUEUES=12 -- --frames-in-flight 6
run fif6_q16 GPU_MAX_HW_QUEUES=16 -- --frames-in-flight 6
run fif3 X=1 -- --frames-in-flight 3
run base X=1 --
