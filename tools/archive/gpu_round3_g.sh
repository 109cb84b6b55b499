#!/bin/bash
# round 3, call G: which sort serves four frames in flight better (the wide downsweeps hold 72 KB of LDS and 8 waves per workgroup)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2 3; do
  for S in ws lsd8; do
    MSPLAT_SORT=$S timeout 300 python bench.py --steps 500 --warmup 50 --no-cpu-baseline --serial-frames 8 --profile-frames 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg2 sort $S  fps %.0f' % d['value'])"
  done
done
for S in ws lsd8; do
  MSPLAT_SORT=$S timeout 600 python bench.py --workload cfg3 --steps 100 --warmup 20 --prewarm 50 --no-cpu-baseline --serial-frames 8 --profile-frames 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 sort $S  fps %.0f' % d['value'])"
  MSPLAT_SORT=$S timeout 600 python bench.py --workload cfg5 --steps 200 --warmup 20 --prewarm 50 --no-cpu-baseline --serial-frames 8 --profile-frames 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 sort $S  fps %.0f' % d['value'])"
done
