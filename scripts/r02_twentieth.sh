#!/bin/bash
# wave | workgroup kernel at the ends of the size range
mkdir -p gpurun_out/r02t
O=gpurun_out/r02t
for n in 512 768 1280; do
EPH_FORCE=wave python scripts/time_path.py $n 5000 0 >> $O/time.log 2>&1
EPH_FORCE=wg python scripts/time_path.py $n 5000 0 >> $O/time.log 2>&1
done
for n in 24576 32768 65536; do
EPH_FORCE=wave python scripts/time_path.py $n 200 0 >> $O/time.log 2>&1
EPH_FORCE=wg python scripts/time_path.py $n 200 0 >> $O/time.log 2>&1
done
cat $O/time.log
