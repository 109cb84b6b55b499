#!/bin/bash
# r4 run t: two-pass variants (gate items per thread, grid of the grid-stride projection), serial stage times + frames in flight
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
for wl in cfg4 cfg3s; do
for v in base gate8 gate4 pg2k pg8k pg16k base; do
  MSPLAT_LIB_PATH=$PWD/tools/bin/variants/libmsplat_$v.so timeout 300 python bench.py --workload $wl --steps 100 --warmup 30 --prewarm 60 --serial-frames 64 --no-cpu-baseline --profile-frames 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['serial']['stages_ms']
print('$wl %-6s: %5.0f frames/s in flight | serial %.4f ms  project %.4f binning %.4f composite %.4f' % ('$v', d['value'], d['serial']['ms_per_frame'], s['project'], s['binning'], s['composite']))"
done
done
