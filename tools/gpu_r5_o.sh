#!/bin/bash
# round 5, run o: is LDS what the projection and the compositor fight over?  project_kernel's one-wave workgroups hold 17 KB of LDS
# each, nine of them fill a CU's 160 KB and leave room for two compositor waves.  Variant library: extra dynamic LDS per projection
# workgroup (MSPLAT_X_PROJ_LDS) so that only 6 / 5 / 4 / 3 fit a CU; serial + frames in flight
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {  # label, env..., -- args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 200 python bench.py --no-cpu-baseline --gpus 1 --serial-frames 64 "$@" 2>gpurun_out/err_o.txt | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); s=d.get('serial',{})
    print('%-18s fps %.0f   serial %.4f ms  serial project %.1f us   in flight: project %.3f composite_kernel %.3f sort %.3f binning %.3f' % ('$label', d['value'], s.get('ms_per_frame', 0), 1e3 * s['stages_ms']['project'], d['stages_ms']['project'], d['stages_ms']['composite_kernel'], d['stages_ms']['sort_total'], d['stages_ms']['binning']))
except Exception as e:
    print('$label', 'FAILED', e); print(open('gpurun_out/err_o.txt').read()[-600:])"
}
V=$PWD/tools/bin/variants/libmsplat_projlds.so
run base X=1 -- --steps 20 --warmup 5
for x in 0 9200 14600 22500 35900; do
  run projlds_$x MSPLAT_LIB_PATH=$V MSPLAT_X_PROJ_LDS=$x -- --steps 20 --warmup 5
done
run base X=1 -- --steps 20 --warmup 5
for x in 14600 22500; do
  run projlds_${x}_100 MSPLAT_LIB_PATH=$V MSPLAT_X_PROJ_LDS=$x -- --steps 100 --warmup 20
done
run base_100 X=1 -- --steps 100 --warmup 20
