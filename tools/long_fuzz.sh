#!/bin/bash
# Run ON THE GPU BOX (gpurun): the randomised parity suite with many seeds; the log (HEAD, seed count, pass / fail counts, duration) goes
# to gpurun_out/<tag>.txt and is committed under profiles/.   Usage: bash tools/long_fuzz.sh <seeds> <tag> <head>
set -u
SEEDS=${1:-2000}; TAG=${2:-r03_fuzz}; HEAD=${3:-unknown}
cd "${GRAFT_REPO_ROOT:?run via gpurun}"
OUT=gpurun_out/$TAG.txt
{
  echo "long fuzz run: F1P_FUZZ_SEEDS=$SEEDS  tree=$HEAD (+ uncommitted changes at run time, if any)  $(date -u +%Y-%m-%dT%H:%M:%SZ)"
  echo "tests: tests/test_gpu_fuzz.py (lattice: 3 schedules + shards + occupancy rules vs the oracle; kmpc: f32 filter / generated controls vs the oracle; footprint; stmpc: f32 filter + time-parallel decision vs the all-fp64 kernel and the oracle)"
  F1P_FUZZ_SEEDS=$SEEDS python -m pytest tests/test_gpu_fuzz.py -q -p no:cacheprovider 2>&1 | tail -15
} > $OUT 2>&1
tail -5 $OUT
