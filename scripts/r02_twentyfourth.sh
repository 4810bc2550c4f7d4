#!/bin/bash
# pair side alone (layout 5) with ablations: 1 no ds_write, 2 no loop barrier, 4 no loads in the loop
mkdir -p gpurun_out/r02x
O=gpurun_out/r02x
python scripts/time_path.py 4096 10000 0 2>&1 >> $O/time.log
for v in s1 s1a1 s1a2 s1a4 s1a3 s1a7; do
EPH_AMD_LIBRARY=$PWD/ephemeris_explorer_amd/libephemeris_amd_exp_$v.so python scripts/time_path.py 4096 10000 0 2>&1 >> $O/time.log
done
cat $O/time.log
