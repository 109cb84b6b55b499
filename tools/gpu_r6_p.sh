#!/bin/bash
# round 6, run p: every context's stream on its own share of the CUs (hipExtStreamCreateWithCUMask): G = 2 (contexts 0, 2 | 1, 3) and
# G = 4 (a quarter of every XCD's CUs each) against the shared GPU.  variant build -DMSPLAT_X_CU_MASK, same box, alternating
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
for rep in 1 2; do
  for G in 1 2 4; do
    MSPLAT_X_CU_GROUPS=$G MSPLAT_LIB_PATH=$V/libmsplat_cumask.so timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 5 --serial-frames 32 2>>gpurun_out/r06p_err.txt | fps "groups_$G"
  done
done
for pool in 640 2560; do
  MSPLAT_X_CU_GROUPS=4 MSPLAT_LIB_PATH=$V/libmsplat_cumask.so timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 5 --serial-frames 32 --compositor-waves $pool 2>>gpurun_out/r06p_err.txt | fps "groups_4_pool$pool"
  MSPLAT_X_CU_GROUPS=2 MSPLAT_LIB_PATH=$V/libmsplat_cumask.so timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 5 --serial-frames 32 --compositor-waves $pool 2>>gpurun_out/r06p_err.txt | fps "groups_2_pool$pool"
done
tail -3 gpurun_out/r06p_err.txt
