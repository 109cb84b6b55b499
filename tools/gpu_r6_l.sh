#!/bin/bash
# round 6, run l: fewer streams with fatter launches?  Two 1920x1080 views of one Sort in ONE chain (cfg2v2) with 1 / 2 / 3 / 4 frames in
# flight against the single-view frame with 1 / 2 / 3 / 4 in flight (views per second is what compares)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
fps() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); v=d['config']['views']
        print('%-16s %.0f frames/s = %.0f views/s  (%.4f ms per frame)' % ('$1', d['value'], v*d['value'], d['ms_per_step']))
"; }
for P in 1 2 3 4; do
  timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 --serial-frames 8 --profile-frames 1 --frames-in-flight $P 2>>gpurun_out/r06l_err.txt | fps "mono_P$P"
  timeout 300 python bench.py --no-cpu-baseline --workload cfg2v2 --steps 200 --warmup 20 --serial-frames 8 --profile-frames 1 --frames-in-flight $P 2>>gpurun_out/r06l_err.txt | fps "two_views_P$P"
done
tail -2 gpurun_out/r06l_err.txt
