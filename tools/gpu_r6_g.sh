#!/bin/bash
# round 6, run g: where the two-ranks-on-one-device bench spends its time; the tail pool A/B (driver protocol, same box)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
( time MSPLAT_BENCH_VERBOSE=1 MSPLAT_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --no-cpu-baseline --also "" --steps 6 --warmup 2 --prewarm 24 --serial-frames 8 --profile-frames 1 ) > gpurun_out/r06g_two_ranks.log 2>&1
grep "\[bench" gpurun_out/r06g_two_ranks.log | head -40; tail -4 gpurun_out/r06g_two_ranks.log | cut -c1-300
fps() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); b=d['block_ms']
        print('$1 fps %.0f  block ms median %.4f min %.4f p90 %.4f  serial %.4f ms' % (d['value'], b['median'], b['min'], b['p90'], d['serial']['ms_per_frame']))
"; }
for rep in 1 2 3; do
  MSPLAT_X_TAILPOOL=0 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --serial-frames 32 2>>gpurun_out/r06g_err.txt | fps "tailpool_off"
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --serial-frames 32 2>>gpurun_out/r06g_err.txt | fps "tailpool_on "
done
for wl in cfg3 cfg5; do
  MSPLAT_X_TAILPOOL=0 timeout 300 python bench.py --no-cpu-baseline --workload $wl --steps 20 --warmup 5 --serial-frames 16 --prewarm 100 2>>gpurun_out/r06g_err.txt | fps "${wl}_tailpool_off"
  timeout 300 python bench.py --no-cpu-baseline --workload $wl --steps 20 --warmup 5 --serial-frames 16 --prewarm 100 2>>gpurun_out/r06g_err.txt | fps "${wl}_tailpool_on "
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "frames_in_flight or scheduling_modes or async" -p no:cacheprovider 2>&1 | tail -3
tail -3 gpurun_out/r06g_err.txt
