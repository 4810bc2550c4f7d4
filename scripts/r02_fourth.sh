#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests/test_gpu_plot.py tests/test_gpu_cli.py tests/test_gpu_fast.py -m gpu -q ) > $OUT/pytest_new.log 2>&1
tail -25 $OUT/pytest_new.log
for LAY in 1 3; do EPH_WG_LAYOUT=$LAY python scripts/time_path.py 4096 3000 0; done > $OUT/layout_time.log 2>&1
for LAY in 1 3; do EPH_WG_LAYOUT=$LAY python scripts/time_path.py 5000 1000 0; done >> $OUT/layout_time.log 2>&1
cat $OUT/layout_time.log
( EPH_WG_LAYOUT=3 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_horizon.py tests/test_gpu_shard.py -m gpu -q -k "accel or plummer or full_size or ranks_on_one" ) > $OUT/pytest_layout3.log 2>&1
tail -8 $OUT/pytest_layout3.log
