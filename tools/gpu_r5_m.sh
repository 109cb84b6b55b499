#!/bin/bash
# round 5, run m: final evidence of the tree as committed -- GPU suite, long soak (4 x 20000 frames), two-pass fuzz (600 cases, seed 5),
# the driver's bench line
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -rs ) > gpurun_out/r05_final_gpu_tests.log 2>&1
grep -E "passed|failed|SKIPPED" gpurun_out/r05_final_gpu_tests.log | cut -c1-200
( time timeout 900 python tools/soak.py --frames 20000 ) > gpurun_out/r05_soak_long.log 2>&1
tail -6 gpurun_out/r05_soak_long.log
( time timeout 900 python tools/two_pass_fuzz.py --cases 600 --seed 5 ) > gpurun_out/r05_two_pass_fuzz.log 2>&1
tail -4 gpurun_out/r05_two_pass_fuzz.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_final_bench_steps20.json 2> gpurun_out/r05_final_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r05_final_bench_steps20.json')); print('fps', d['value'], 'serial', d['serial']['ms_per_frame'], 'roof', d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'], d.get('two_pass_check'))"
