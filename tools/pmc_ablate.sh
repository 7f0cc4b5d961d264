#!/bin/bash
# Run ON THE GPU BOX (gpurun): per-wave VALU / SALU / LDS / transcendental instruction counts, busy and wait cycles and the kernel
# time of the filter kernel for the ablated builds libf1p_ab<N>.so (make LIB=libf1p_abN.so OBJDIR=build_abN EXTRA=-DF1P_MIX_ABLATE=N;
# bits: 1 no f32 candidate evaluation, 2 no look-ahead scans, 4 no nearest scan, 8 no station loop).  Usage: pmc_ablate.sh [N ...]
set -eu
cd "${GRAFT_REPO_ROOT:?run via gpurun}"; export TMPDIR=/tmp
LIST=${*:-0 8 1 6 7 15}
for a in $LIST; do
  if [ $a = 0 ]; then unset F1P_LIBRARY; else export F1P_LIBRARY=$PWD/f1tenth_planning_amd/csrc/libf1p_ab$a.so; fi
  python3 tools/time_mixed.py > gpurun_out/abl_$a.time 2>&1 || true
  rm -rf gpurun_out/abl_$a gpurun_out/ablb_$a
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM -f csv -d gpurun_out/abl_$a -o run -- python3 tools/time_mixed.py > gpurun_out/abl_$a.log 2>&1 || true
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE -f csv -d gpurun_out/ablb_$a -o run -- python3 tools/time_mixed.py > gpurun_out/ablb_$a.log 2>&1 || true
  python3 - <<PY
import csv,glob,collections
rows=[]
for d in ("abl_$a","ablb_$a"):
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True): rows+=list(csv.DictReader(open(f)))
agg=collections.defaultdict(list)
for r in rows:
    if "lattice_filter" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
t=[l.strip() for l in open("gpurun_out/abl_$a.time") if "filter" in l or " ms" in l]
print("ablate $a:", " | ".join(t))
dur=collections.defaultdict(list)
for f in glob.glob("gpurun_out/abl_$a/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): dur[r["Kernel_Name"].split("(")[0][-28:]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("    kernel us (under the profiler):", {k: round(sum(v)/len(v), 1) for k,v in dur.items() if "lattice" in k})
print("    per wave:", {k: round(sum(v)/len(v)/16384, 1) for k,v in sorted(agg.items())})
PY
done
