#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/r04d; mkdir -p $OUT
export TMPDIR=/tmp
( echo "== default"; EPH_TRACE_SUMMARY=1 python scripts/time_sweep_parts2.py 262144 fresh 2>&1
  echo "== shared stream"; EPH_CRAFT_SHARED_STREAM=1 EPH_TRACE_SUMMARY=1 python scripts/time_sweep_parts2.py 262144 fresh 2>&1
  echo "== HSA_SCRATCH_SINGLE_LIMIT big"; HSA_SCRATCH_SINGLE_LIMIT=4294967296 EPH_TRACE_SUMMARY=1 python scripts/time_sweep_parts2.py 262144 fresh 2>&1
  echo "== HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0"; HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0 EPH_TRACE_SUMMARY=1 python scripts/time_sweep_parts2.py 262144 fresh 2>&1
  echo "== OCC 1 (no spills?)"; EPH_CRAFT_OCC=1 EPH_TRACE_SUMMARY=1 python scripts/time_sweep_parts2.py 262144 fresh 2>&1
) > $OUT/sweep_parts.txt 2>&1
grep -v "^summary" $OUT/sweep_parts.txt
