#!/bin/bash
# round 5, run j: compositor variant that leaves finished strip PAIRS out of the inner loop (chosen per batch from `alive`;
# three copies of the loop, occupancy bound 6 to stay at 80 VGPRs) against the tree: parity subset, then every workload
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
V=$PWD/tools/bin/variants/libmsplat_deadpairs.so
MSPLAT_LIB_PATH=$V timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_shaders.py -m gpu -q --timeout 300 -p no:cacheprovider -x -k "config2 or config1 or two_pass or in_flight or bands or reference or stereo" 2>&1 | tail -5
run() {  # label, env..., -- args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 200 python bench.py --no-cpu-baseline --gpus 1 --serial-frames 32 "$@" 2>gpurun_out/err_j.txt | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); s=d.get('serial',{})
    print('%-22s fps %.0f   serial %.4f ms  serial compositor kernel %.1f us  in-flight %s' % ('$label', d['value'], s.get('ms_per_frame', 0), 1e3 * d['roofline']['avg_launch_ms'], {k: round(v, 3) for k, v in d.get('stages_ms', {}).items() if k in ('composite_kernel',)}))
except Exception as e:
    print('$label', 'FAILED', e); print(open('gpurun_out/err_j.txt').read()[-600:])"
}
for rep in 1 2; do
run base_cfg2 X=1 -- --steps 20 --warmup 5
run dead_cfg2 MSPLAT_LIB_PATH=$V -- --steps 20 --warmup 5
done
for wl in cfg3 cfg4 cfg5 cfg3s; do
run base_$wl X=1 -- --workload $wl --steps 60 --warmup 20
run dead_$wl MSPLAT_LIB_PATH=$V -- --workload $wl --steps 60 --warmup 20
done
