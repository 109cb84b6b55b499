#!/bin/bash
# r4: full GPU suite after the header split / 8-bit mip levels / point-shader fixtures, then a default bench line
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r4j_smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r4j_smoke.log
timeout 1500 python -m pytest tests -m gpu -q -rP -x > gpurun_out/r4j_gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4j_gpu_tests.log
grep -E "passed|failed|HIP point renderer|HIP vs reference|worst relative" gpurun_out/r4j_gpu_tests.log | tail -30
tail -3 gpurun_out/r4j_smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r4j_bench_default.json 2> gpurun_out/r4j_bench_default.err
python - <<'PY'
import json
for l in open('gpurun_out/r4j_bench_default.json'):
    if l.startswith('{'):
        d=json.loads(l); print('bench', d['value'], d['serial']['ms_per_frame'], d['roofline']['frac'], d['cpu_baseline']['value'])
PY
