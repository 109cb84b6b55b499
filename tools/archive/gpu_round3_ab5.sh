#!/bin/bash
# same-box A/B: round-2 tree vs this tree with the 256-thread three-pass sort for frames in flight; parity tests first
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -k "sort or wide or parity or in_flight or frames" 2>&1 | tail -5 > gpurun_out/r03ab5_tests.log
cat gpurun_out/r03ab5_tests.log
run() {  # label, dir, env, args
  label=$1; dir=$2; envs=$3; shift; shift; shift
  (cd $dir && env $envs timeout 300 python bench.py --no-cpu-baseline --serial-frames 32 --profile-frames 1 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-16s fps %.0f   serial %.4f ms' % ('$label', d['value'], d['serial']['ms_per_frame']))")
}
for rep in 1 2 3; do
  run r2_500 build/r2tree X=1 --steps 500 --warmup 50
  run r3_ws256_500 . X=1 --steps 500 --warmup 50
  run r3_lsd8_500 . MSPLAT_SORT=lsd8 --steps 500 --warmup 50
  run r3_ws512_500 . MSPLAT_WS_THREADS=512 --steps 500 --warmup 50
  run r3_ws256c_500 . MSPLAT_TILE_TABLE=counts --steps 500 --warmup 50
  run r2_steps20 build/r2tree X=1 --steps 20 --warmup 5
  run r3_steps20 . X=1 --steps 20 --warmup 5
done
for w in cfg3 cfg3s; do
  run r3_${w}_ws256 . X=1 --workload $w --steps 200 --warmup 20
  run r3_${w}_lsd8 . MSPLAT_SORT=lsd8 --workload $w --steps 200 --warmup 20
done
