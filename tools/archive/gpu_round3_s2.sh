#!/bin/bash
# frames in flight: XCD-contiguous row pass (MSPLAT_XCD_MAP bit 2) and the counts-based bin offsets, driver protocol and long blocks
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
run() {  # label, env, args
  label=$1; envs=$2; shift; shift
  env $envs timeout 300 python bench.py --no-cpu-baseline --serial-frames 8 --profile-frames 1 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-26s fps %.0f' % ('$label', d['value']))"
}
for rep in 1 2 3; do
  run default_steps20 X=1 --steps 20 --warmup 5
  run xcd1_steps20 MSPLAT_XCD_MAP=1 --steps 20 --warmup 5
  run count_steps20 MSPLAT_TILE_TABLE=count --steps 20 --warmup 5
  run default_500 X=1 --steps 500 --warmup 50
  run xcd1_500 MSPLAT_XCD_MAP=1 --steps 500 --warmup 50
  run count_500 MSPLAT_TILE_TABLE=count --steps 500 --warmup 50
done
