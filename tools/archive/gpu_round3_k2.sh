#!/bin/bash
# compositor pool sweep with the round-3 kernel selection for frames in flight (driver protocol and 500-frame blocks)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
run() {  # label, env, args
  label=$1; envs=$2; shift; shift
  env $envs timeout 300 python bench.py --no-cpu-baseline --serial-frames 8 --profile-frames 1 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-22s fps %.0f' % ('$label', d['value']))"
}
for rep in 1 2; do
for Wv in 768 1024 1280 1536 2048; do
  run pool${Wv}_steps20 MSPLAT_COMP_WAVES=$Wv --steps 20 --warmup 5
  run pool${Wv}_500 MSPLAT_COMP_WAVES=$Wv --steps 500 --warmup 50
done
done
