#!/bin/bash
# layout 5: default, pair side alone, chain side alone (compiler schedule / pinned tile); each with EPH_WG_LAYOUT 5 and 3
mkdir -p gpurun_out/r02w
O=gpurun_out/r02w
for l in 5 3; do
EPH_WG_LAYOUT=$l python scripts/time_path.py 4096 10000 0 2>&1 >> $O/time.log
for v in side1 side2 side2asm; do
EPH_WG_LAYOUT=$l EPH_AMD_LIBRARY=$PWD/ephemeris_explorer_amd/libephemeris_amd_exp_$v.so python scripts/time_path.py 4096 10000 0 2>&1 >> $O/time.log
done
done
cat $O/time.log
