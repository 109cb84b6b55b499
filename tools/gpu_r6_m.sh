#!/bin/bash
# round 6, run m: runtime knobs that touch launch latency -- HIP_FORCE_DEV_KERNARG (kernel arguments in device memory), driver protocol + serial
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
fps() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['serial']
        print('%-22s %.0f frames/s (20-frame blocks)  serial %.4f ms  latency %.4f ms' % ('$1', d['value'], s['ms_per_frame'], s['single_frame_latency_ms_host_to_host']))
"; }
for rep in 1 2; do
  for v in unset 0 1; do
    if [ $v = unset ]; then timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --serial-frames 64 2>>gpurun_out/r06m_err.txt | fps "kernarg_default"
    else HIP_FORCE_DEV_KERNARG=$v timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --serial-frames 64 2>>gpurun_out/r06m_err.txt | fps "HIP_FORCE_DEV_KERNARG=$v"; fi
  done
done
tail -2 gpurun_out/r06m_err.txt
