#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
run() {  # label, dir, args
  label=$1; dir=$2; shift; shift
  (cd $dir && timeout 300 python bench.py --no-cpu-baseline --serial-frames 32 --profile-frames 1 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-16s fps %.0f   serial %.4f ms' % ('$label', d['value'], d['serial']['ms_per_frame']))")
}
for rep in 1 2 3; do
  run r2_500 build/r2tree --steps 500 --warmup 50
  run head_500 . --steps 500 --warmup 50
  run nobatch_500 build/t_nobatch --steps 500 --warmup 50
done
