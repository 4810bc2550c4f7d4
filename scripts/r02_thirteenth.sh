#!/bin/bash
# pair-wave loop variants (EPH_PAIR_LOOP 0..3), all with the seeded reciprocal
mkdir -p gpurun_out/r02m
O=gpurun_out/r02m
for i in 1 2; do
for v in loop0 loop1 loop2 loop3; do
EPH_AMD_LIBRARY=$PWD/ephemeris_explorer_amd/libephemeris_amd_exp_$v.so python scripts/time_path.py 4096 20000 0 >> $O/time.log 2>&1
done
done
cat $O/time.log
