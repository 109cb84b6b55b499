#!/bin/bash
# round 5, run e: GPU suite (full-size two-pass parity, async overflow warning, priorities made permanent) + the driver's bench line
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x 2>&1 | tail -12
for rep in 1 2; do
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5e_bench_$rep.json
python -c "
import json; d=json.load(open('gpurun_out/r5e_bench_$rep.json')); print('fps', d['value'], 'serial', d['serial']['ms_per_frame'], d.get('two_pass_check'))"
done
timeout 300 python bench.py --gpus 1 --workload cfg3 --steps 60 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 fps', d['value'], 'serial', d['serial']['ms_per_frame'], d.get('two_pass_check'))"
