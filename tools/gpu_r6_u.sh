#!/bin/bash
# round 6, run u: what do the stage events (msplat_config.enable_timing = 8: every 8th frame of a context) cost the timed region?
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
fps() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)
        print('%-30s in flight %.4f ms/frame (%.0f fps)' % ('$1', d['ms_per_step'], d['value']))
"; }
run() { local label=$1; shift; timeout 300 python bench.py --no-cpu-baseline --warmup 5 --serial-frames 8 "$@" 2>>gpurun_out/r06u_err.txt | fps "$label"; }
for rep in 1 2 3; do
  run "stride 8 (default), 20" --steps 20
  run "stride 0, 20" --steps 20 --timing-stride 0
  run "stride 64, 20" --steps 20 --timing-stride 64
done
run "stride 8 (default), 200" --steps 200
run "stride 0, 200" --steps 200 --timing-stride 0
tail -3 gpurun_out/r06u_err.txt
