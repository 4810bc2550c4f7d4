#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( time timeout 2400 python -m pytest tests -m gpu -q --durations=8 ) > $OUT/pytest.log 2>&1
tail -20 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
bash scripts/round_profile.sh r02 > $OUT/round_profile.log 2>&1
tail -5 $OUT/round_profile.log
for L in 1 3; do EPH_WG_LAYOUT=$L EPH_DEBUG_WG=4 python scripts/wg_cycles.py 4096; done > $OUT/wg_cycles.log 2>&1; cat $OUT/wg_cycles.log
scripts/sample_clocks.sh $OUT/clocks_default.csv python scripts/time_path.py 4096 60000 0 > $OUT/time_default.log 2>&1; cat $OUT/time_default.log
awk -F, '{print $9, $13}' $OUT/clocks_default.csv | sort | uniq -c | sort -rn | head -4
