#!/bin/bash
# layout 5 and 3: both sides, pair side alone (EPH_DEBUG_WG=1), chain side alone (=2); results of 1 / 2 are meaningless
mkdir -p gpurun_out/r02v
O=gpurun_out/r02v
for l in 5 3; do
for d in 0 1 2; do
EPH_WG_LAYOUT=$l EPH_DEBUG_WG=$d python scripts/time_path.py 4096 10000 0 2>&1 | grep -v "max .dpos" >> $O/time.log
done
done
cat $O/time.log
