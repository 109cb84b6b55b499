#!/bin/bash
# frames in flight: compositor wave pool sweep (default bench command, 500-frame blocks)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
for W in ${POOLS:-1024 768 1280 1536 2048 1024}; do
  MSPLAT_COMP_WAVES=$W timeout 300 python bench.py --steps 500 --warmup 50 --no-cpu-baseline --serial-frames 8 --profile-frames 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pool $W  fps %.0f  overlapped compk %.4f' % (d['value'], d['stages_ms'].get('composite_kernel', 0)))"
done
