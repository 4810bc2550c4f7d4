#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02k
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
scripts/ubench/chain2 > $OUT/chain2.log 2>&1; cat $OUT/chain2.log
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plot.py -m gpu -q -k "tile_count or plot" ) > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log | head -2
