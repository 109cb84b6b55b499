#!/bin/bash
# round 6, run f: where the GPU suite's 12 minutes go (durations), then the default bench line of this tree
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --durations=40 ) > gpurun_out/r06f_gpu_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r06f_gpu_tests.log | tail -3; grep -A45 "slowest 40" gpurun_out/r06f_gpu_tests.log | cut -c1-150
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r06f_bench_steps20.json 2>gpurun_out/r06f_err.txt
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06f_bench_steps20.json') if l.startswith('{')][-1])
print("fps %.0f timed_blocks %d timed_seconds %.2f  valu %s" % (d['value'], d['timed_blocks'], d['timed_seconds'], json.dumps(d['roofline']['valu'])))
print("cpu_baseline", d.get('cpu_baseline',{}).get('value'))
PY
tail -3 gpurun_out/r06f_err.txt
