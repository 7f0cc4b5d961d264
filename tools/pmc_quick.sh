#!/bin/bash
# Run ON THE GPU BOX from a tree's root: bash tools/pmc_quick.sh <tag> [bench args]  -- kernel trace + ONE PMC pass of the timed region (A/B of two trees)
set -u
TAG=${1:-q}; [ $# -gt 0 ] && shift
OUT=$(pwd)/gpurun_out
ARGS="--steps 60 --warmup 10 --no-cpu-baseline --no-secondary --latency-iters 0 --only-timed $*"
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -f csv -d $OUT/${TAG}_trace -o run -- python3 bench.py $ARGS > $OUT/${TAG}_trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM -f csv -d $OUT/${TAG}_pmc_a -o run -- python3 bench.py $ARGS > $OUT/${TAG}_pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv -d $OUT/${TAG}_pmc_b -o run -- python3 bench.py $ARGS > $OUT/${TAG}_pmc_b.log 2>&1
python3 tools/prof_summary.py $OUT/${TAG}_trace $OUT/${TAG}_pmc_a $OUT/${TAG}_pmc_b > $OUT/${TAG}_summary.md 2>$OUT/${TAG}_summary.err
grep -E "k_lattice" $OUT/${TAG}_summary.md | cut -c1-400
