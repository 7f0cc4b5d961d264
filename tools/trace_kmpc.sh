#!/bin/bash
# per-kernel durations of the kmpc workload (GPU box): bash tools/trace_kmpc.sh <lib.so> <egos>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export F1P_LIBRARY=$ROOT/f1tenth_planning_amd/csrc/$1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trk; rocprofv3 --kernel-trace --stats -f csv -d /tmp/trk -o run -- python3 $ROOT/bench.py --workload kmpc --egos $2 --steps 10 --warmup 2 --no-cpu-baseline > /tmp/trk.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob('/tmp/trk/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:40], r['Calls'], 'avg_us %.1f' % (float(r['AverageNs'])/1e3), 'min %.1f' % (float(r['MinNs'])/1e3))
PY
