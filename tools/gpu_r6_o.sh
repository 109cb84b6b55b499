#!/bin/bash
# round 6, run o: ablation -- project_kernel reading the records in STORAGE order (slot = rank: wrong pixels) instead of gathering them by
# sorted index: what would a projection that does not follow the sort's order gain?  same box, alternating
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
fps() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['serial']; st=s['stages_ms']; f=d['stages_ms']
        print('%-14s in flight %.4f ms/frame (%.0f fps)  serial %.4f ms | serial us: sort %.1f project %.1f binning %.1f compk %.1f | in flight us: sort %.0f project %.0f binning %.0f compk %.0f' % ('$1', d['ms_per_step'], d['value'], s['ms_per_frame'], 1e3*st['sort_total'], 1e3*st['project'], 1e3*st['binning'], 1e3*st['composite_kernel'], 1e3*f['sort_total'], 1e3*f['project'], 1e3*f['binning'], 1e3*f['composite_kernel']))
"; }
V=$PWD/tools/bin/variants
for wl in cfg2 cfg3 cfg5; do
  for rep in 1 2; do
    timeout 300 python bench.py --no-cpu-baseline --workload $wl --steps 200 --warmup 5 --serial-frames 64 --two-pass off 2>>gpurun_out/r06o_err.txt | fps "product_$wl"
    MSPLAT_LIB_PATH=$V/libmsplat_projident.so timeout 300 python bench.py --no-cpu-baseline --workload $wl --steps 200 --warmup 5 --serial-frames 64 --two-pass off 2>>gpurun_out/r06o_err.txt | fps "identity_$wl"
  done
done
tail -3 gpurun_out/r06o_err.txt
