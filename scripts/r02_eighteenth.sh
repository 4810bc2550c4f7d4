#!/bin/bash
# force-only kernel (SRKN methods, start-up) and Stormer13 under layouts 3 / 5 / 6
mkdir -p gpurun_out/r02r
O=gpurun_out/r02r
for l in 3 5 6; do
EPH_WG_LAYOUT=$l python scripts/time_path.py 4096 1000 0 BlanesMoan14A >> $O/time.log 2>&1
EPH_WG_LAYOUT=$l python scripts/time_path.py 4096 10000 0 Stormer13 >> $O/time.log 2>&1
EPH_WG_LAYOUT=$l python scripts/time_path.py 4096 10000 0 >> $O/time.log 2>&1
done
EPH_WG_LAYOUT=5 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_horizon.py tests/test_gpu_shard.py -x -q > $O/pytest5.log 2>&1
head -2 $O/pytest5.log
cat $O/time.log
