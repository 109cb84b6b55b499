#!/bin/bash
# which round-3 commit cost the throughput mode its 5 %?  same box, 4 frames in flight, 8-bit sort everywhere (MSPLAT_SORT=lsd8)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
run() {  # label, dir
  (cd $2 && MSPLAT_SORT=lsd8 timeout 300 python bench.py --no-cpu-baseline --serial-frames 16 --profile-frames 1 --steps 500 --warmup 50 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-16s fps %.0f   serial %.4f ms' % ('$1', d['value'], d['serial']['ms_per_frame']))")
}
for rep in 1 2; do
  run r2 build/r2tree
  for c in b75bb59 9b5261d b89c1a3 1ed4c64 c8d8e1c c0908eb; do run $c build/t_$c; done
  run head .
done
