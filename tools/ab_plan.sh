#!/bin/bash
# Same-box A/B of libf1p variants on the timed region alone: bash tools/ab_plan.sh [lib ...]   (paths relative to csrc/; run from the tree's root on the GPU box)
# Alternates the variants REPS times (default 3): box-to-box spread is +-1.5 us per plan, run-to-run on one box +-0.2 us.
ROOT=$(pwd)
REPS=${REPS:-3}
for rep in $(seq $REPS); do
  for L in "$@"; do
    F1P_LIBRARY=$ROOT/f1tenth_planning_amd/csrc/$L python3 $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --latency-iters 0 --full-record /tmp/ab_full.json ${AB_ARGS:-} 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$L', 'ms_per_plan %.5f' % d['ms_per_step'], 'kernels(us): pro %.2f fil %.2f ref %.2f sel %.2f' % tuple(1e3 * (d.get(k) or 0) for k in ('kernel_ms_prologue', 'kernel_ms_filter3', 'kernel_ms_refine', 'kernel_ms_select')), 'audit', d.get('audit_mismatching_egos'))
"
  done
done
