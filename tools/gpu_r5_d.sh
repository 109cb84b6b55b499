#!/bin/bash
# round 5, run d: wave issue priorities of the chain kernels (c), project_kernel (p) and the compositor's heaviest items (m = its
# highest level), variants built with -DMSPLAT_CHAIN_PRIO / -DMSPLAT_PROJ_PRIO / -DMSPLAT_COMP_PRIO_MAX; driver protocol + 500-frame blocks
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {  # label, env..., -- args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 120 python bench.py --no-cpu-baseline --gpus 1 --serial-frames 16 "$@" 2>gpurun_out/err_d.txt | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); s=d.get('serial',{})
    print('%-26s fps %.0f   serial %.4f ms  in-flight stages %s' % ('$label', d['value'], s.get('ms_per_frame', 0), {k: round(v, 3) for k, v in d.get('stages_ms', {}).items() if k != 'frames_averaged'}))
except Exception as e:
    print('$label', 'FAILED', e); print(open('gpurun_out/err_d.txt').read()[-600:])"
}
V=$PWD/tools/bin/variants
for rep in 1 2; do
run base X=1 -- --steps 20 --warmup 5
for v in c3m2 c3m0 c3m3 c2m1 c3p1m0 c3p2m1 c3p2m2 c3p3m2; do
  run $v MSPLAT_LIB_PATH=$V/libmsplat_$v.so -- --steps 20 --warmup 5
done
done
run base_500 X=1 -- --steps 500 --warmup 50
for v in c3m2 c3m0 c3p2m1 c3p3m2; do
  run ${v}_500 MSPLAT_LIB_PATH=$V/libmsplat_$v.so -- --steps 500 --warmup 50
done
for wl in cfg3 cfg4 cfg5 cfg3s; do
  run base_$wl X=1 -- --workload $wl --steps 60 --warmup 20
  run c3m2_$wl MSPLAT_LIB_PATH=$V/libmsplat_c3m2.so -- --workload $wl --steps 60 --warmup 20
done
