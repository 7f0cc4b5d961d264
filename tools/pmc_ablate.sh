#!/bin/bash
# Run ON THE GPU BOX from the tree's root: bash tools/pmc_ablate.sh <tag> lib [lib ...]   (libs relative to csrc/)
# Per library variant: rocprofv3 kernel trace (durations) + one PMC pass (SQ_INSTS_VALU, SQ_WAVES, SQ_BUSY_CYCLES, SQ_ACTIVE_INST_VALU) of the timed region
# alone; prints per lattice kernel: average us, wave-instructions, instructions per wave.  Measurement variants (F1P_F3_ABLATE) are NOT valid plans.
set -u
TAG=${1:-abl}; [ $# -gt 0 ] && shift
OUT=$(pwd)/gpurun_out
export TMPDIR=/tmp
mkdir -p $OUT
ARGS="--steps 60 --warmup 10 --no-cpu-baseline --no-secondary --latency-iters 0 --only-timed --full-record /tmp/abl_full.json"
for L in "$@"; do
  N=$(basename $L .so)
  export F1P_LIBRARY=$(pwd)/f1tenth_planning_amd/csrc/$L
  rocprofv3 --kernel-trace -f csv -d $OUT/${TAG}_${N}_trace -o run -- python3 bench.py $ARGS > $OUT/${TAG}_${N}_trace.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS -f csv -d $OUT/${TAG}_${N}_pmc -o run -- python3 bench.py $ARGS > $OUT/${TAG}_${N}_pmc.log 2>&1
  python3 - $OUT/${TAG}_${N}_trace $OUT/${TAG}_${N}_pmc $N <<'PY'
import csv, glob, os, sys
from collections import defaultdict
tr, pm, name = sys.argv[1:4]
dur = defaultdict(list)
for f in glob.glob(os.path.join(tr, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
cnt = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(pm, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(dur):
    if "lattice" not in k:
        continue
    d = sorted(dur[k]); d = d[len(d) // 10: len(d) - len(d) // 10] or d
    c = {n: sum(v) / len(v) for n, v in cnt.get(k, {}).items()}
    w = c.get("SQ_WAVES", 0) or 1
    short = k.split("(")[0].replace("void f1p::", "")
    print(f"{name:<12} {short:<46} {sum(d) / len(d):8.2f} us  valu {c.get('SQ_INSTS_VALU', 0):12.0f}  per wave {c.get('SQ_INSTS_VALU', 0) / w:8.1f}  salu/wave {c.get('SQ_INSTS_SALU', 0) / w:7.1f}  lds/wave {c.get('SQ_INSTS_LDS', 0) / w:6.1f}  waves {w:7.0f}  active_valu {c.get('SQ_ACTIVE_INST_VALU', 0):12.0f}")
PY
done
