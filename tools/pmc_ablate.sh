#!/bin/bash
# Run ON THE GPU BOX: VALU / SALU / LDS / transcendental instruction counts of k_lattice_filter for the ablated builds libf1p_ab<N>.so
set -eu
cd "${GRAFT_REPO_ROOT:?run via gpurun}"; export TMPDIR=/tmp
for a in 0 8 1 3 7; do
  if [ $a = 0 ]; then unset F1P_LIBRARY; else export F1P_LIBRARY=$PWD/f1tenth_planning_amd/csrc/libf1p_ab$a.so; fi
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS -f csv -d gpurun_out/abl_$a -o run -- python3 tools/time_mixed.py > gpurun_out/abl_$a.log 2>&1
  python3 - <<PY
import csv,glob,collections
rows=[]
for f in glob.glob("gpurun_out/abl_$a/**/*counter_collection.csv", recursive=True): rows+=list(csv.DictReader(open(f)))
agg=collections.defaultdict(list)
for r in rows:
    if "lattice_filter" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("ablate $a", {k: round(sum(v)/len(v)/16384) for k,v in agg.items()}, "per wave")
PY
done
