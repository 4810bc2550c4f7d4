#!/bin/bash
# layout 5 with the tail work on its idle wave, layout 6 (late history), default 3
mkdir -p gpurun_out/r02q
O=gpurun_out/r02q
for i in 1 2; do
for l in 3 5 6; do
EPH_WG_LAYOUT=$l python scripts/time_path.py 4096 20000 0 >> $O/time.log 2>&1
done
done
for l in 5 6; do
for n in 2048 3000 6000 8000; do
EPH_WG_LAYOUT=$l python scripts/time_path.py $n 5000 0 >> $O/time.log 2>&1
done
done
EPH_WG_LAYOUT=5 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest5.log 2>&1
head -2 $O/pytest5.log
cat $O/time.log
