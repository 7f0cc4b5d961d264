#!/bin/bash
# A/B of the STREAMED shooting kernel: tools/ab_kmpc_stream.sh libA.so libB.so ...   (E = 1024 and 8192, two runs each)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for L in "$@"; do
  for E in 1024 8192; do
    F1P_LIBRARY=$ROOT/f1tenth_planning_amd/csrc/$L python3 $ROOT/bench.py --workload kmpc --kmpc-stream --egos $E --steps 200 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', 'E=$E', 'ms %.4f' % d['roofline']['kernel_ms'], 'GB/s %.0f' % d['roofline']['achieved'], 'mism', d.get('parity'))"
  done
done
done
