#!/bin/bash
# A/B a set of libf1p variants on the GPU box: bash tools/ab_bench.sh [lib ...]   (paths relative to csrc/)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for L in "$@"; do
  for rep in 1 2; do
    F1P_LIBRARY=$ROOT/f1tenth_planning_amd/csrc/$L python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --latency-iters 0 2>&1 | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$L', 'value %.4g' % d['value'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'blocked', d['blocked_egos'])
    elif 'rror' in line: print(line.strip())
"
  done
done
