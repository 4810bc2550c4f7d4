#!/bin/bash
# sixteen-wave layouts 7 / 8 and the tail-wave hand-off of 5 / 6 against the default (3)
mkdir -p gpurun_out/r02p
O=gpurun_out/r02p
for i in 1 2; do
for l in 3 5 6 7 8; do
EPH_WG_LAYOUT=$l python scripts/time_path.py 4096 20000 0 >> $O/time.log 2>&1
done
done
for l in 6 7 8; do
EPH_WG_LAYOUT=$l timeout 600 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest$l.log 2>&1
head -2 $O/pytest$l.log
done
cat $O/time.log
