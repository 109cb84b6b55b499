#!/bin/bash
# same-box A/B of the round-2 tree (build/r2tree = commit f98a1fc) against this tree, throughput and serial modes
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
export TMPDIR=/tmp
run() {  # label, dir, env..., -- args
  label=$1; dir=$2; shift; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  (cd $dir && env "${envs[@]}" timeout 300 python bench.py --no-cpu-baseline --serial-frames 64 --profile-frames 1 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-26s fps %.0f   serial %.4f ms   enqueue %.4f ms' % ('$label', d['value'], d['serial']['ms_per_frame'], d['host_enqueue_ms_per_frame']))")
}
for rep in 1 2; do
  run r2_500 build/r2tree -- --steps 500 --warmup 50
  run r3_500 . -- --steps 500 --warmup 50
  run r3_500_search . MSPLAT_TILE_TABLE=search -- --steps 500 --warmup 50
  run r3_500_nosplit . MSPLAT_HEAVY_SPLIT=0 -- --steps 500 --warmup 50
  run r2_steps20 build/r2tree -- --steps 20 --warmup 5
  run r3_steps20 . -- --steps 20 --warmup 5
done
run r3_500_ws . MSPLAT_SORT=ws -- --steps 500 --warmup 50
run r3_500_fused8k . MSPLAT_FUSED_MAX_CHUNKS=0 -- --steps 500 --warmup 50
