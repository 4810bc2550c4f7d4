#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests/test_gpu_plot.py tests/test_gpu_cli.py tests/test_gpu_fast.py -m gpu -q ) > $OUT/pytest_new.log 2>&1
tail -12 $OUT/pytest_new.log
for U in 4 8; do for S in 32 64; do EPH_FAST_UNROLL=$U EPH_FAST_SLICES=$S python scripts/time_path.py 4096 2000 4; done; done > $OUT/fast_time.log 2>&1
cat $OUT/fast_time.log
python scripts/time_path.py 4096 3000 0 > $OUT/default_time.log 2>&1; cat $OUT/default_time.log
for N in 2048 3072 8192 16384; do for F in wave wg; do EPH_FORCE=$F python scripts/time_path.py $N 300 0; done; done > $OUT/crossover.log 2>&1
cat $OUT/crossover.log
