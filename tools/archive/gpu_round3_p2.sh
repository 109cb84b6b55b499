#!/bin/bash
# cull upsweep over 8192-key chunks in two halves (76 VGPRs instead of 126): parity + kernel durations at 6 M
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "large_cloud or sort or cfg3 or cfg4 or scene" 2>&1 | tail -3
bash tools/gpu_round3_m2.sh --workload cfg3 2>&1 | grep -v "^\[" | head -14
