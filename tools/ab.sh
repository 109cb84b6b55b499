#!/bin/bash
# usage: tools/ab.sh [bench args]: prints fps for one-frame-at-a-time and for the default frames in flight
cd ${GRAFT_REPO_ROOT:-.}
for P in 1 4; do
  for rep in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --frames-in-flight $P "$@" 2>&1 | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('P=$P fps %.1f ms %.4f lat %.4f compk %.4f serial-compk %s' % (d['value'], d['ms_per_step'], d['single_frame_latency_ms'], d['stages_ms']['composite_kernel'], d['roofline']['avg_launch_ms_one_frame_at_a_time']))
"
  done
done
