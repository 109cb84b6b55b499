#!/bin/bash
# round 6, run c: is the WORKGROUP DISPATCHER what frames in flight compete for?  project_kernel launches 15.6 k one-wave workgroups
# per frame; MSPLAT_X_PROJ_GRID caps its grid (grid-stride, persistent workgroups).  Driver protocol, same box, alternating.
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
fps() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d.get('serial',{})
        print('$1 fps %.0f  ms/step %.4f  serial %.4f ms  in flight us: sort %.0f project %.0f binning %.0f composite_kernel %.0f | serial us: sort %.0f project %.0f binning %.0f comp %.0f' % (d['value'], d['ms_per_step'], s.get('ms_per_frame',0), 1e3*d['stages_ms']['sort_total'], 1e3*d['stages_ms']['project'], 1e3*d['stages_ms']['binning'], 1e3*d['stages_ms']['composite_kernel'], 1e3*s['stages_ms']['sort_total'], 1e3*s['stages_ms']['project'], 1e3*s['stages_ms']['binning'], 1e3*s['stages_ms']['composite_kernel']))
"; }
for rep in 1 2; do
  for g in 0 1152 2304 4608 9216; do
    MSPLAT_X_PROJ_GRID=$g timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --serial-frames 64 2>>gpurun_out/r06c_err.txt | fps "projgrid_$g"
  done
done
MSPLAT_LIB_PATH=$PWD/tools/bin/variants/libmsplat_stamps.so timeout 900 python tools/stamp_timeline.py 4 160 > gpurun_out/r06c_stamp_timeline.txt 2>>gpurun_out/r06c_err.txt
cat gpurun_out/r06c_stamp_timeline.txt | cut -c1-220
MSPLAT_X_PROJ_GRID=2304 MSPLAT_LIB_PATH=$PWD/tools/bin/variants/libmsplat_stamps.so timeout 900 python tools/stamp_timeline.py 4 160 > gpurun_out/r06c_stamp_timeline_projgrid2304.txt 2>>gpurun_out/r06c_err.txt
cat gpurun_out/r06c_stamp_timeline_projgrid2304.txt | cut -c1-220
tail -3 gpurun_out/r06c_err.txt
