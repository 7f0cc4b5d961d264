#!/bin/bash
# host-boundary plan() latency for several builds: tools/ab_latency.sh libA.so libB.so ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for L in "$@"; do
  F1P_LIBRARY=$ROOT/f1tenth_planning_amd/csrc/$L python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); l=d['plan_latency_host_boundary']; print('$L', 'kernel %.4f' % d['roofline']['kernel_ms'], 'pinned p50 %.4f p95 %.4f' % (l['p50_ms'], l['p95_ms']), 'pageable p50 %.4f' % l['pageable_host_arrays']['p50_ms'], 'no-traj p50 %.4f' % l['without_best_traj']['p50_ms'], 'bnb p50 %.4f' % l['branch_and_bound']['p50_ms'])"
done
