#!/bin/bash
# 6 M splats with frames in flight: the four 8-bit passes (the size rule's choice) against three passes in both workgroup forms
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
run() {  # label, env, args
  label=$1; envs=$2; shift; shift
  env $envs timeout 300 python bench.py --no-cpu-baseline --serial-frames 8 --profile-frames 1 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-26s fps %.0f' % ('$label', d['value']))"
}
for rep in 1 2; do
for w in cfg3 cfg3s cfg4; do
  run ${w}_lsd8 X=1 --workload $w --steps 100 --warmup 10
  run ${w}_wide256 MSPLAT_SORT=wide --workload $w --steps 100 --warmup 10
  run ${w}_wide512 "MSPLAT_SORT=wide MSPLAT_WS_THREADS=512" --workload $w --steps 100 --warmup 10
done
done
