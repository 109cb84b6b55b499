#!/bin/bash
# round 6, run n: project_kernel's LDS staging by quarters (5 KB per one-wave workgroup instead of 17 KB): tools/ubench_launch.hip says
# the 17-KB workgroups starve the other frames' multi-wave workgroups of LDS.  product (4 parts) against 8 parts (9 KB), 16 parts
# (17 KB, wave-scope sync) and the tree before the change ("head"), same box, alternating; then the parity subset.
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
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "config2 or frames_in_flight or test_image or smoke or stereo or two_pass" -p no:cacheprovider 2>&1 | tail -3
for steps in 20 200; do
  for rep in 1 2; do
    timeout 300 python bench.py --no-cpu-baseline --steps $steps --warmup 5 --serial-frames 64 2>>gpurun_out/r06n_err.txt | fps "parts4_$steps"
    for w in parts8 parts16 head; do
      MSPLAT_LIB_PATH=$V/libmsplat_$w.so timeout 300 python bench.py --no-cpu-baseline --steps $steps --warmup 5 --serial-frames 64 2>>gpurun_out/r06n_err.txt | fps "${w}_$steps"
    done
  done
done
for wl in cfg3 cfg5; do
  timeout 300 python bench.py --no-cpu-baseline --workload $wl --steps 20 --warmup 5 --serial-frames 32 2>>gpurun_out/r06n_err.txt | fps "parts4_$wl"
  MSPLAT_LIB_PATH=$V/libmsplat_head.so timeout 300 python bench.py --no-cpu-baseline --workload $wl --steps 20 --warmup 5 --serial-frames 32 2>>gpurun_out/r06n_err.txt | fps "head_$wl"
done
tail -3 gpurun_out/r06n_err.txt
