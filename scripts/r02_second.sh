#!/bin/bash
# Round 2, second GPU call: whole -m gpu suite (no -x), fast path as two launches, workgroup kernel role layouts.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( time timeout 2400 python -m pytest tests -m gpu -q --durations=12 ) > $OUT/pytest.log 2>&1
tail -25 $OUT/pytest.log
for S in 8 16 32; do EPH_FAST_SLICES=$S python scripts/time_path.py 4096 2000 4; done > $OUT/fast_time.log 2>&1
python scripts/time_path.py 1024 2000 4 >> $OUT/fast_time.log 2>&1
python scripts/time_path.py 16384 200 4 >> $OUT/fast_time.log 2>&1
python scripts/time_path.py 65536 20 4 >> $OUT/fast_time.log 2>&1
cat $OUT/fast_time.log
for LAY in 0 1; do for DBG in 0 8; do
  EPH_WG_LAYOUT=$LAY EPH_DEBUG_WG=$DBG python scripts/time_path.py 4096 3000 0
done; done > $OUT/layout_time.log 2>&1
cat $OUT/layout_time.log
for LAY in 0 1; do EPH_WG_LAYOUT=$LAY EPH_DEBUG_WG=4 python scripts/wg_cycles.py 4096; EPH_WG_LAYOUT=$LAY EPH_DEBUG_WG=12 python scripts/wg_cycles.py 4096; done > $OUT/wg_cycles.log 2>&1
cat $OUT/wg_cycles.log
EPH_WG_LAYOUT=1 scripts/sample_clocks.sh $OUT/clocks_layout1.csv python scripts/time_path.py 4096 60000 0 > $OUT/time_layout1.log 2>&1
cat $OUT/time_layout1.log; awk -F, '{print $9, $13}' $OUT/clocks_layout1.csv | sort | uniq -c | sort -rn | head -5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o fast_stats -- python $GRAFT_REPO_ROOT/scripts/time_path.py 4096 2000 4 > $OUT/fast_stats.log 2>&1
head -8 $OUT/fast_stats_kernel_stats.csv
