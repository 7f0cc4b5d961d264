#!/bin/bash
# A/B of the K1-class kernels: tools/ab_k1.sh libA.so libB.so ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for L in "$@"; do
  F1P_LIBRARY=$ROOT/f1tenth_planning_amd/csrc/$L python3 $ROOT/bench.py --workload pursuit --steps 30 --warmup 3 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', 'kernel ms %.4f' % d['kernel_ms'], 'plans/s %.3g' % d['value'], d.get('parity'), 'cpu %.3g' % d['cpu_baseline']['value'])"
done
