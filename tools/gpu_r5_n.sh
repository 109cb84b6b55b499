#!/bin/bash
# round 5, run n: the compositor's staging phase (the dependent chain between two batches of a tile's list) at wave priority 2 / 3,
# the inner loop back at the item's own (tools/archive/r5_compositor_staging_priority.patch); serial + frames in flight
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {  # label, env..., -- args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 200 python bench.py --no-cpu-baseline --gpus 1 --serial-frames 64 "$@" 2>gpurun_out/err_n.txt | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); s=d.get('serial',{})
    print('%-18s fps %.0f   serial %.4f ms  serial compositor kernel %.1f us  in flight %.3f' % ('$label', d['value'], s.get('ms_per_frame', 0), 1e3 * d['roofline']['avg_launch_ms'], d['stages_ms']['composite_kernel']))
except Exception as e:
    print('$label', 'FAILED', e); print(open('gpurun_out/err_n.txt').read()[-600:])"
}
V=$PWD/tools/bin/variants
for rep in 1 2; do
run base_cfg2 X=1 -- --steps 20 --warmup 5
run stage2_cfg2 MSPLAT_LIB_PATH=$V/libmsplat_stage2.so -- --steps 20 --warmup 5
run stage3_cfg2 MSPLAT_LIB_PATH=$V/libmsplat_stage3.so -- --steps 20 --warmup 5
done
for wl in cfg4 cfg5 cfg3; do
run base_$wl X=1 -- --workload $wl --steps 60 --warmup 20
run stage2_$wl MSPLAT_LIB_PATH=$V/libmsplat_stage2.so -- --workload $wl --steps 60 --warmup 20
run stage3_$wl MSPLAT_LIB_PATH=$V/libmsplat_stage3.so -- --workload $wl --steps 60 --warmup 20
done
