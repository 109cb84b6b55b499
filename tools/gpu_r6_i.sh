#!/bin/bash
# (project_kernel_mw is in commit 8eef668: tools/build_variant.sh projw4 -DMSPLAT_X_PROJ_WAVES=4)
# round 6, run i: the plain projection as workgroups of 2 / 4 / 8 waves (same per-wave work): does a stream of one-wave workgroups
# starve the other frames' multi-wave workgroups?  product build against the variants, same box, alternating
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
fps() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['serial']; st=s['stages_ms']; f=d['stages_ms']
        print('%-12s in flight %.4f ms/frame (%.0f fps)  serial %.4f ms | serial us: sort %.1f project %.1f binning %.1f compk %.1f | in flight us: sort %.0f project %.0f binning %.0f compk %.0f' % ('$1', d['ms_per_step'], d['value'], s['ms_per_frame'], 1e3*st['sort_total'], 1e3*st['project'], 1e3*st['binning'], 1e3*st['composite_kernel'], 1e3*f['sort_total'], 1e3*f['project'], 1e3*f['binning'], 1e3*f['composite_kernel']))
"; }
V=$PWD/tools/bin/variants
for steps in 20 200; do
  for rep in 1 2; do
    timeout 300 python bench.py --no-cpu-baseline --steps $steps --warmup 5 --serial-frames 64 2>>gpurun_out/r06i_err.txt | fps "product_$steps"
    for w in 2 4 8; do
      MSPLAT_LIB_PATH=$V/libmsplat_projw$w.so timeout 300 python bench.py --no-cpu-baseline --steps $steps --warmup 5 --serial-frames 64 2>>gpurun_out/r06i_err.txt | fps "projw${w}_$steps"
    done
  done
done
MSPLAT_LIB_PATH=$V/libmsplat_projw4.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "config2 or frames_in_flight or test_image or smoke" -p no:cacheprovider 2>&1 | tail -3
tail -3 gpurun_out/r06i_err.txt
