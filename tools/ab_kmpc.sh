#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for L in "$@"; do
  F1P_LIBRARY=$ROOT/f1tenth_planning_amd/csrc/$L python3 $ROOT/bench.py --workload kmpc --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', 'ms %.4f' % d['roofline']['kernel_ms'], 'GB/s %.0f' % d['roofline']['achieved'])"
done
