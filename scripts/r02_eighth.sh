#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02h
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shard.py tests/test_gpu_variants.py -m gpu -q -k "not million and not 1e5" ) > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
python scripts/wg_cycles.py 4096 > $OUT/srkn_time.log 2>&1; head -1 $OUT/srkn_time.log
python scripts/wg_cycles.py 1024 >> $OUT/srkn_time.log 2>&1; python scripts/wg_cycles.py 300 >> $OUT/srkn_time.log 2>&1; grep Blanes $OUT/srkn_time.log
