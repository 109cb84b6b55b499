#!/bin/bash
# round 6, run e: the GPU suite on the tree with the spill fixes / whole-frame parity / weighted bands, then the same-box A/B of
# round 5's library against this tree (driver protocol, serial, and the 6 M workloads)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rs ) > gpurun_out/r06e_gpu_tests.log 2>&1
tail -15 gpurun_out/r06e_gpu_tests.log; grep "_check_whole_frame" gpurun_out/r06e_gpu_tests.log | head
fps() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d.get('serial',{})
        print('$1 fps %.0f  ms/step %.4f  serial %.4f ms  | serial us: sort %.1f project %.1f binning %.1f comp %.1f' % (d['value'], d['ms_per_step'], s.get('ms_per_frame',0), 1e3*s['stages_ms']['sort_total'], 1e3*s['stages_ms']['project'], 1e3*s['stages_ms']['binning'], 1e3*s['stages_ms']['composite_kernel']))
"; }
R5=$PWD/tools/bin/variants/libmsplat_r5.so
for rep in 1 2 3; do
  MSPLAT_LIB_PATH=$R5 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --serial-frames 64 2>>gpurun_out/r06e_err.txt | fps "r5_cfg2"
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --serial-frames 64 2>>gpurun_out/r06e_err.txt | fps "r6_cfg2"
done
for wl in cfg3 cfg4 cfg5; do
  MSPLAT_LIB_PATH=$R5 timeout 400 python bench.py --no-cpu-baseline --workload $wl --steps 20 --warmup 5 --serial-frames 32 --prewarm 100 2>>gpurun_out/r06e_err.txt | fps "r5_$wl"
  timeout 400 python bench.py --no-cpu-baseline --workload $wl --steps 20 --warmup 5 --serial-frames 32 --prewarm 100 2>>gpurun_out/r06e_err.txt | fps "r6_$wl"
done
tail -3 gpurun_out/r06e_err.txt
