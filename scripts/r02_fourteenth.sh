#!/bin/bash
# own-tile override in the pair waves: parity, then timing against the build without it (exp_loop0)
mkdir -p gpurun_out/r02n
O=gpurun_out/r02n
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_horizon.py tests/test_gpu_shard.py -x -q > $O/pytest.log 2>&1
for i in 1 2; do
python scripts/time_path.py 4096 20000 0 >> $O/time.log 2>&1
EPH_AMD_LIBRARY=$PWD/ephemeris_explorer_amd/libephemeris_amd_exp_loop0.so python scripts/time_path.py 4096 20000 0 >> $O/time.log 2>&1
done
python scripts/time_path.py 2048 20000 0 >> $O/time.log 2>&1
python scripts/time_path.py 3000 20000 0 >> $O/time.log 2>&1
head -3 $O/pytest.log; cat $O/time.log
