#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/r04i; mkdir -p $OUT
L=$PWD/ephemeris_explorer_amd
for rep in 1 2; do
  for lib in libephemeris_amd.so "$@"; do
    python scripts/ab_step.py $L/$lib
  done
done 2>&1 | tee -a $OUT/ab.txt
