#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02j
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python scripts/time_path.py 4096 3000 0 > $OUT/time.log 2>&1
EPH_DEBUG_WG=16 python scripts/time_path.py 4096 3000 0 >> $OUT/time.log 2>&1
python scripts/time_path.py 3072 3000 0 >> $OUT/time.log 2>&1; python scripts/time_path.py 2048 3000 0 >> $OUT/time.log 2>&1; cat $OUT/time.log
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_horizon.py tests/test_gpu_shard.py -m gpu -q -k "accel or plummer or full_size or ranks_on_one or config5" ) > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log | head -2
python scripts/step_span.py > $OUT/step_span.log 2>&1; head -2 $OUT/step_span.log
