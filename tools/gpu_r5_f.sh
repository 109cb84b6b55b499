#!/bin/bash
# round 5, run f: RCCL exchange behind the C ABI (one-rank loopback, group switch), header split, bench contract tests
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 240 -p no:cacheprovider -x -k "rccl or device_group or pair_overflow" 2>&1 | tail -15
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x 2>&1 | tail -8
