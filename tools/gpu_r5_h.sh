#!/bin/bash
# round 5, run h: the round's evidence set (tools/gpu_profiles.sh r05) + one more A/B: the in-flight sort of the 6 M workloads
# (four 8-bit passes, r3's choice) against the three wide passes now that the chain kernels run at priority 3
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
bash tools/gpu_profiles.sh r05 2>&1 | tail -150
for wl in cfg3 cfg4; do
for v in default wide; do
  if [ $v = wide ]; then export MSPLAT_SORT=wide; else unset MSPLAT_SORT; fi
  timeout 300 python bench.py --workload $wl --steps 60 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$wl sort=$v fps %.0f serial %.4f' % (d['value'], d['serial']['ms_per_frame']))"
done
done
