#!/bin/bash
# round 6, run q: which partition of the CUs?  (variant build -DMSPLAT_X_CU_MASK; 200-frame blocks and the driver's 20-frame blocks)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
fps() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); f=d['stages_ms']
        print('%-28s in flight %.4f ms/frame (%.0f fps) | in flight us: sort %.0f project %.0f binning %.0f compk %.0f' % ('$1', d['ms_per_step'], d['value'], 1e3*f['sort_total'], 1e3*f['project'], 1e3*f['binning'], 1e3*f['composite_kernel']))
"; }
V=$PWD/tools/bin/variants
run() { # label, env...
  local label=$1; shift
  env "$@" MSPLAT_LIB_PATH=$V/libmsplat_cumask.so timeout 300 python bench.py --no-cpu-baseline --steps ${STEPS:-200} --warmup 5 --serial-frames 8 ${EXTRA} 2>>gpurun_out/r06q_err.txt | fps "$label"
}
for rep in 1 2; do
  run shared MSPLAT_X_CU_GROUPS=1
  run "2 mod (0,2|1,3)" MSPLAT_X_CU_GROUPS=2
  run "2 mod (0,1|2,3)" MSPLAT_X_CU_GROUPS=2 MSPLAT_X_CU_PAIR=1
  run "2 block (0,2|1,3)" MSPLAT_X_CU_GROUPS=2 MSPLAT_X_CU_KIND=block
  run "2 se (0,2|1,3)" MSPLAT_X_CU_GROUPS=2 MSPLAT_X_CU_KIND=se
  run "4 soft (3/4 each)" MSPLAT_X_CU_GROUPS=4 MSPLAT_X_CU_KIND=soft
done
STEPS=20 run "shared, 20-frame blocks" MSPLAT_X_CU_GROUPS=1
STEPS=20 run "2 mod, 20-frame blocks" MSPLAT_X_CU_GROUPS=2
STEPS=20 run "shared, 20-frame blocks" MSPLAT_X_CU_GROUPS=1
STEPS=20 run "2 mod, 20-frame blocks" MSPLAT_X_CU_GROUPS=2
for wl in cfg3 cfg4 cfg5 cfg3s; do
  EXTRA="--workload $wl" run "shared $wl" MSPLAT_X_CU_GROUPS=1
  EXTRA="--workload $wl" run "2 mod $wl" MSPLAT_X_CU_GROUPS=2
done
EXTRA="--frames-in-flight 6" run "2 mod, 6 in flight" MSPLAT_X_CU_GROUPS=2
EXTRA="--frames-in-flight 3" run "2 mod, 3 in flight" MSPLAT_X_CU_GROUPS=2
tail -3 gpurun_out/r06q_err.txt
