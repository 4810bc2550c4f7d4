#!/bin/bash
# default layout 5: force-only kernel under 3 / 5 / 6 (BlanesMoan14A, wall time), and the wave | workgroup crossover sizes
mkdir -p gpurun_out/r02s
O=gpurun_out/r02s
for l in 3 5 6; do
EPH_WG_LAYOUT=$l python scripts/time_path.py 4096 1000 0 BlanesMoan14A >> $O/time.log 2>&1
done
for n in 1024 1536 2048 8192 12288 16384; do
EPH_FORCE=wave python scripts/time_path.py $n 3000 0 >> $O/time.log 2>&1
EPH_FORCE=wg python scripts/time_path.py $n 3000 0 >> $O/time.log 2>&1
done
cat $O/time.log
