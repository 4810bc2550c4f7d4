#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02g
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for LAY in 3 4; do EPH_WG_LAYOUT=$LAY python scripts/time_path.py 4096 3000 0; done > $OUT/layout_time.log 2>&1; cat $OUT/layout_time.log
( EPH_WG_LAYOUT=4 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "accel or plummer" ) > $OUT/pytest_layout4.log 2>&1; tail -3 $OUT/pytest_layout4.log
python bench.py --workload craft --steps 3 > $OUT/bench_craft.json 2> $OUT/bench_craft.err; cut -c1-400 $OUT/bench_craft.json; tail -2 $OUT/bench_craft.err
python scripts/craft_30d.py 20000 3 > $OUT/craft_small.json 2> $OUT/craft_small.err; cat $OUT/craft_small.json; tail -3 $OUT/craft_small.err
timeout 900 python scripts/craft_30d.py 1000000 30 > $OUT/craft_30d.json 2> $OUT/craft_30d.err; cat $OUT/craft_30d.json; tail -3 $OUT/craft_30d.err
