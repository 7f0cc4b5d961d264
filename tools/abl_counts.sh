#!/bin/bash
# VALU instruction counts of ablation builds (GPU box): bash tools/abl_counts.sh lib1.so lib2.so ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  export F1P_LIBRARY=$ROOT/f1tenth_planning_amd/csrc/$L
  rm -rf /tmp/ablc; rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -f csv -d /tmp/ablc -o run -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --latency-iters 0 > /tmp/ablc.log 2>&1
  python3 - "$L" <<'PY'
import csv, glob, sys
from collections import defaultdict
agg = defaultdict(list)
for f in glob.glob('/tmp/ablc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_lattice' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
w = sum(agg['SQ_WAVES'])/max(len(agg['SQ_WAVES']),1)
print(sys.argv[1], {k: round(sum(v)/len(v)/w, 1) for k, v in agg.items() if k != 'SQ_WAVES'}, 'per wave; waves', w)
PY
done
