#!/bin/bash
# round 6, run r: msplat_config.cu_partition in the product: tests, then bench.py --cu-partition auto against off (driver protocol and
# 200-frame blocks, alternating), compositor pools under the partition, the other workloads
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
fps() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); f=d['stages_ms']
        print('%-30s in flight %.4f ms/frame (%.0f fps) %s | in flight us: sort %.0f project %.0f binning %.0f compk %.0f' % ('$1', d['ms_per_step'], d['value'], d['config']['cu_partition'], 1e3*f['sort_total'], 1e3*f['project'], 1e3*f['binning'], 1e3*f['composite_kernel']))
"; }
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "cu_partition or frames_in_flight or frame_modes or async or stereo or smoke" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_bench_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -3
run() { local label=$1; shift; timeout 300 python bench.py --no-cpu-baseline --warmup 5 --serial-frames 8 "$@" 2>>gpurun_out/r06r_err.txt | fps "$label"; }
for rep in 1 2 3; do
  run "off, 20-frame blocks" --steps 20 --cu-partition off
  run "auto, 20-frame blocks" --steps 20
done
for rep in 1 2; do
  run "off, 200-frame blocks" --steps 200 --cu-partition off
  run "auto, 200-frame blocks" --steps 200
done
for pool in 896 1024 1536 1792 2048; do
  run "auto, pool $pool, 20" --steps 20 --compositor-waves $pool
  run "auto, pool $pool, 200" --steps 200 --compositor-waves $pool
done
for wl in cfg3 cfg4 cfg5 cfg3s; do
  run "off $wl" --steps 20 --workload $wl --cu-partition off
  run "auto $wl" --steps 20 --workload $wl
done
tail -3 gpurun_out/r06r_err.txt
