#!/bin/bash
# per-kernel durations of the lattice workload (GPU box): bash tools/trace_lattice.sh <lib.so> [bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export F1P_LIBRARY=$ROOT/f1tenth_planning_amd/csrc/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trl; rocprofv3 --kernel-trace --stats -f csv -d /tmp/trl -o run -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --latency-iters 0 "$@" > /tmp/trl.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob('/tmp/trl/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:70], r['Calls'], 'avg_us %.1f' % (float(r['AverageNs'])/1e3), 'min %.1f' % (float(r['MinNs'])/1e3))
PY
