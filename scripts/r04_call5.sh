#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/r04e; mkdir -p $OUT
( for t in summary sleep step none; do echo "== PRE=$t"; PRE=$t python scripts/time_sweep_parts2.py 262144 fresh 2>&1; done
) > $OUT/sweep_parts7.txt 2>&1
grep -v "^summary" $OUT/sweep_parts7.txt | grep -v "^propagate 3[0-9].* summary [012]\."
