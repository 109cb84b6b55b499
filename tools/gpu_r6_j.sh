#!/bin/bash
# round 6, run j: final-tree validation -- smoke, the whole GPU suite, the driver's command, the 2 / 3-rank CPU-side control flow is in the CPU suite
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rs ) > gpurun_out/r06j_gpu_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/r06j_gpu_tests.log | tail -3; grep "^real" gpurun_out/r06j_gpu_tests.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r06j_bench_steps20.json 2> gpurun_out/r06j_bench_steps20.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06j_bench_steps20.json') if l.startswith('{')][-1])
r=d['roofline']
print("value %.0f frames/s  ms_per_step %.4f  timed %d blocks %.2f s  roofline frac %.3f traffic %s (%s)  valu frac %.3f useful %.3f  cpu_baseline %.1f (%s cores)" % (d['value'], d['ms_per_step'], d['timed_blocks'], d['timed_seconds'], r['frac'], r['traffic'], (r['traffic_source'] or '')[:40], r['valu']['frac_of_fp32_vector_peak'], r['valu']['useful_eval_frac'], d['cpu_baseline']['value'], d['cpu_baseline']['cores']))
PY
grep "^real" gpurun_out/r06j_bench_steps20.err
