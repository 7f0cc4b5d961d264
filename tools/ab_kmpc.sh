#!/bin/bash
# A/B of the shooting kernel: tools/ab_kmpc.sh libA.so libB.so ...   (E = 1024 and 8192)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for L in "$@"; do
  for E in 1024 8192; do
    F1P_LIBRARY=$ROOT/f1tenth_planning_amd/csrc/$L python3 $ROOT/bench.py --workload kmpc --egos $E --steps 200 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', 'E=$E', 'ms %.4f' % d['roofline']['kernel_ms'], 'GB/s %.0f' % d['roofline']['achieved'])"
  done
done
