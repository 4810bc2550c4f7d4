#!/bin/bash
# workgroups of 8 / 4 bodies for <= 2048 / <= 1024 targets: parity, then timing against 16-body workgroups and the wave form
mkdir -p gpurun_out/r02ab
O=gpurun_out/r02ab
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shard.py -x -q > $O/pytest.log 2>&1
head -3 $O/pytest.log
for n in 512 768 1024 1536 2048; do
python scripts/time_path.py $n 10000 0 2>&1 | tail -1 >> $O/time.log
EPH_WG_BODIES=16 python scripts/time_path.py $n 10000 0 2>&1 | tail -1 >> $O/time.log
EPH_FORCE=wg python scripts/time_path.py $n 10000 0 2>&1 | tail -1 >> $O/time.log
EPH_FORCE=wave python scripts/time_path.py $n 10000 0 2>&1 | tail -1 >> $O/time.log
done
EPH_WG_BODIES=8 python scripts/time_path.py 1024 10000 0 2>&1 | tail -1 >> $O/time.log
EPH_WG_BODIES=8 EPH_FORCE=wg python scripts/time_path.py 512 10000 0 2>&1 | tail -1 >> $O/time.log
EPH_WG_BODIES=4 EPH_FORCE=wg python scripts/time_path.py 256 10000 0 2>&1 | tail -1 >> $O/time.log
EPH_FORCE=wave python scripts/time_path.py 256 10000 0 2>&1 | tail -1 >> $O/time.log
cat $O/time.log
