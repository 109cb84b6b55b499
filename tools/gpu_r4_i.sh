#!/bin/bash
# round 4, run i: full GPU suite after removing the experiments; cfg5 line with both eyes in one chain; default line
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
T=r4i
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x ) > gpurun_out/${T}_gpu_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/${T}_gpu_tests.log | tail -3
grep -E "^(FAILED|ERROR)|^E  " gpurun_out/${T}_gpu_tests.log | head -30
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "stereo or frames_in_flight_bit" 2>&1 | tail -3
timeout 600 python bench.py --workload cfg5 --steps 200 --warmup 20 --prewarm 100 --no-cpu-baseline > gpurun_out/${T}_cfg5.json 2> gpurun_out/${T}_cfg5.err
python - <<PY
import json
d = json.load(open("gpurun_out/${T}_cfg5.json"))
print("cfg5: value %.0f  serial %.4f ms  frac %.3f  stages %s" % (d["value"], d["serial"]["ms_per_frame"], d["roofline"]["frac"], {k: (round(v["us"], 1), round(v["frac"], 3)) for k, v in d["roofline"]["stages"].items() if isinstance(v, dict)}))
print(d["roofline"]["bytes_definition"])
PY
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_cfg2_steps20.json 2> gpurun_out/${T}_cfg2_steps20.err
python -c "import json; d=json.load(open('gpurun_out/${T}_cfg2_steps20.json')); print('cfg2 steps20: %.0f  serial %.4f' % (d['value'], d['serial']['ms_per_frame']))"
