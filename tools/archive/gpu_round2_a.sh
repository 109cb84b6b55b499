#!/bin/bash
# GPU call A of round 2: full GPU test suite (new config 3/4/5 gates), VALU microbenchmark, compositor probe dump,
# and the restructured bench in both modes.  Everything lands in gpurun_out/r2a_*.
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --durations=12 ) > gpurun_out/r2a_tests.log 2>&1
tail -25 gpurun_out/r2a_tests.log
timeout 120 tools/bin/ubench_valu > gpurun_out/r2a_ubench.log 2>&1; cat gpurun_out/r2a_ubench.log
timeout 300 python tools/probe_dump.py cfg2 0 16 > gpurun_out/r2a_probe.log 2>&1; tail -6 gpurun_out/r2a_probe.log
timeout 300 python bench.py --steps 200 --warmup 50 --cpu-frames 2 > gpurun_out/r2a_bench_default.json 2> gpurun_out/r2a_bench_default.err; tail -c 3000 gpurun_out/r2a_bench_default.json
timeout 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --frames-in-flight 1 > gpurun_out/r2a_bench_serial.json 2> gpurun_out/r2a_bench_serial.err; tail -c 1500 gpurun_out/r2a_bench_serial.json
timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2a_bench_20.json 2> gpurun_out/r2a_bench_20.err; tail -c 600 gpurun_out/r2a_bench_20.json
