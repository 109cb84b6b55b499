#!/bin/bash
# same-box A/B under the driver's protocol (--steps 20 --warmup 5): round-2 tree vs this tree (cheaper host marshalling, cached poses)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {  # label, dir, env, args
  label=$1; dir=$2; envs=$3; shift; shift; shift
  (cd $dir && env $envs timeout 300 python bench.py --no-cpu-baseline --serial-frames 32 --profile-frames 1 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-18s fps %.0f   serial %.4f ms  host enqueue %.1f us/frame' % ('$label', d['value'], d['serial']['ms_per_frame'], 1e3*d['host_enqueue_ms_per_frame']))")
}
for rep in 1 2 3; do
  run r2_steps20 build/r2tree X=1 --steps 20 --warmup 5
  run r3_steps20 . X=1 --steps 20 --warmup 5
  run r3_steps20_cnt . MSPLAT_TILE_TABLE=counts --steps 20 --warmup 5
  run r3_steps20_fif3 . X=1 --steps 20 --warmup 5 --frames-in-flight 3
  run r2_500 build/r2tree X=1 --steps 500 --warmup 50
  run r3_500 . X=1 --steps 500 --warmup 50
done
