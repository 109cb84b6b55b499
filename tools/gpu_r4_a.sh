#!/bin/bash
# round 4, run a: GPU suite after the prune + bench contract changes (gather_check over gloo, roofline.stages), default bench line
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rP -x ) > gpurun_out/r4a_gpu_tests.log 2>&1
grep -E "passed|failed|error|SKIPPED" gpurun_out/r4a_gpu_tests.log | tail -5
grep -E "projection parity" gpurun_out/r4a_gpu_tests.log | sort -t' ' -k6 -g | tail -3
grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/r4a_gpu_tests.log | head -20
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r4a_cfg2_bench_steps20.json 2> gpurun_out/r4a_cfg2_bench_steps20.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4a_cfg2_bench_steps20.json"))
print("value", d["value"], "serial", d["serial"]["ms_per_frame"], "frac", d["roofline"]["frac"], "moved", d["frame_moved_frac"], d["frame_moved_frac_serial"])
print({k: (round(v["us"], 1), round(v["frac"], 3)) for k, v in d["roofline"]["stages"].items() if isinstance(v, dict)})
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["threads"])
PY
tail -3 gpurun_out/r4a_cfg2_bench_steps20.err
