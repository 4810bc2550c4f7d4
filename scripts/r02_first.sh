#!/bin/bash
# Round 2, first GPU call: the whole -m gpu suite, the fast path's first timings, and the step kernel's evidence
# (cycle accounting of the workgroup kernel + clock / power samples while it runs).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
nproc > $OUT/nproc.txt
rocm-smi --showclocks --showpower --csv > $OUT/smi_idle.csv 2>&1
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 ) > $OUT/pytest.log 2>&1
tail -30 $OUT/pytest.log
for S in 16 32 64; do EPH_FAST_SLICES=$S python scripts/time_path.py 4096 2000 4; done > $OUT/fast_time.log 2>&1
python scripts/time_path.py 16384 200 4 >> $OUT/fast_time.log 2>&1
python scripts/time_path.py 1024 2000 4 >> $OUT/fast_time.log 2>&1
python scripts/time_path.py 65536 20 4 >> $OUT/fast_time.log 2>&1
cat $OUT/fast_time.log
scripts/sample_clocks.sh $OUT/clocks_wg.csv python scripts/time_path.py 4096 60000 0 > $OUT/time_wg.log 2>&1
scripts/sample_clocks.sh $OUT/clocks_fast.csv python scripts/time_path.py 4096 100000 4 > $OUT/time_fast.log 2>&1
cat $OUT/time_wg.log $OUT/time_fast.log
EPH_DEBUG_WG=4 python scripts/wg_cycles.py 4096 > $OUT/wg_cycles.log 2>&1
cat $OUT/wg_cycles.log
head -5 $OUT/clocks_wg.csv; tail -3 $OUT/clocks_wg.csv
