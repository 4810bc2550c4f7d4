#!/bin/bash
# twelve-wave layouts 5 / 6 against the default (3): timing, then parity under the faster one
mkdir -p gpurun_out/r02o
O=gpurun_out/r02o
for i in 1 2; do
for l in 3 5 6; do
EPH_WG_LAYOUT=$l python scripts/time_path.py 4096 20000 0 >> $O/time.log 2>&1
done
done
for l in 5 6; do
EPH_WG_LAYOUT=$l python scripts/time_path.py 2048 20000 0 >> $O/time.log 2>&1
EPH_WG_LAYOUT=$l python scripts/time_path.py 3000 20000 0 >> $O/time.log 2>&1
done
EPH_WG_LAYOUT=5 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_horizon.py -x -q > $O/pytest5.log 2>&1
EPH_WG_LAYOUT=6 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_horizon.py -x -q > $O/pytest6.log 2>&1
head -2 $O/pytest5.log; head -2 $O/pytest6.log; cat $O/time.log
