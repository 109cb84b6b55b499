#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
run() {  # label, dir, args
  label=$1; dir=$2; shift; shift
  (cd $dir && timeout 300 python bench.py --no-cpu-baseline --serial-frames 32 --profile-frames 1 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-16s fps %.0f   serial %.4f ms' % ('$label', d['value'], d['serial']['ms_per_frame']))")
}
for rep in 1 2 3; do
  run r2_500 build/r2tree --steps 500 --warmup 50
  run r3_500 . --steps 500 --warmup 50
  run r2_steps20 build/r2tree --steps 20 --warmup 5
  run r3_steps20 . --steps 20 --warmup 5
done
