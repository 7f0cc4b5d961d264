#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash tools/profile_gpu.sh <tag> <headline-kernel-substring> <schedule> [bench args...]
# (round 5: gfx950 has no SQ_INSTS_VALU_TRANS; the per-class counters are SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_{F32,F64}, _CVT, _INT32, _INT64)
# Kernel trace + separate PMC passes (never combined with other trace domains), outputs under gpurun_out/<tag>_*; the markdown
# summary and the machine-readable <tag>_pmc.json (parsed by bench.py at run time) are what gets copied into profiles/.
set -u
TAG=${1:-r03}; HEAD=${2:-k_lattice_filter3}; SCHED=${3:-mixed}
shift $(( $# < 3 ? $# : 3 ))      # (a bare `shift 3` with fewer arguments shifts nothing and the tag would reach bench.py)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-secondary --latency-iters 0 --only-timed $*"
export TMPDIR=/tmp
cd $ROOT
rocprofv3 --kernel-trace --stats -f csv -d $OUT/${TAG}_trace -o run -- python3 bench.py $ARGS > $OUT/${TAG}_trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64" "GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C -f csv -d $OUT/${TAG}_pmc_$N -o run -- python3 bench.py $ARGS > $OUT/${TAG}_pmc_$N.log 2>&1 || echo "pmc pass $C failed (see log)"
done
# PROF_CONFIG (optional env): the JSON object bench.py's load_pmc() matches the profile against; default = the lattice headline
CONFIG=${PROF_CONFIG:-"{\"egos\": 4096, \"cands\": 256, \"stations\": 50, \"workload\": \"lattice\", \"generator\": \"clothoid\", \"schedule\": \"$SCHED\"}"}
python3 tools/prof_summary.py --json $OUT/${TAG}_pmc.json --headline "$HEAD" \
    --config "$CONFIG" \
    $OUT/${TAG}_trace $OUT/${TAG}_pmc_* > $OUT/${TAG}_summary.md 2>$OUT/${TAG}_summary.err
grep -h '"metric"' $OUT/${TAG}_trace.log | head -1 > $OUT/${TAG}_bench_line.json
head -40 $OUT/${TAG}_summary.md | cut -c1-220
