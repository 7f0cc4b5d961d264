#!/bin/bash
# Run ON THE GPU BOX (gpurun): the randomised parity suite with many seeds; the log (HEAD, seed count, pass / fail counts, duration) goes
# to gpurun_out/<tag>.txt and is committed under profiles/.   Usage: bash tools/long_fuzz.sh <seeds> <tag> <head> [seconds] [first seed]
set -u
SEEDS=${1:-2000}; TAG=${2:-r03_fuzz}; HEAD=${3:-unknown}
cd "${GRAFT_REPO_ROOT:?run via gpurun}"
OUT=gpurun_out/$TAG.txt
{
  echo "long fuzz run: F1P_FUZZ_SEEDS=$SEEDS F1P_FUZZ_SEED0=${5:-0}  tree=$HEAD (+ uncommitted changes at run time, if any)  $(date -u +%Y-%m-%dT%H:%M:%SZ)"
  echo "tests: tests/test_gpu_fuzz.py (lattice: 3 schedules + shards + occupancy rules vs the oracle; kmpc: f32 filter / generated controls vs the oracle; footprint; stmpc: f32 filter + time-parallel decision vs the all-fp64 kernel and the oracle)"
  # the test run writes to a file of its own and is bounded HERE (4th argument, seconds; 2 000 seeds take ~480 s, 10 000 ~ 2 400 s): a
  # `timeout` around this script would kill the pipe's tail with it and leave no record of the cases that did run
  F1P_FUZZ_SEEDS=$SEEDS F1P_FUZZ_SEED0=${5:-0} timeout ${4:-3000} python -m pytest tests/test_gpu_fuzz.py -q -p no:cacheprovider > gpurun_out/$TAG.raw 2>&1
  echo "pytest exit code $? (124 = the time bound cut the run: the lines below are the cases that ran)"
  tail -15 gpurun_out/$TAG.raw; rm -f gpurun_out/$TAG.raw
} > $OUT 2>&1
tail -5 $OUT
