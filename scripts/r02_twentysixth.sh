#!/bin/bash
# LDS writes spread over the pair-wave iteration (EPH_PAIR_EARLY_WRITE) against the default
mkdir -p gpurun_out/r02aa
O=gpurun_out/r02aa
for i in 1 2; do
python scripts/time_path.py 4096 20000 0 >> $O/time.log 2>&1
EPH_AMD_LIBRARY=$PWD/ephemeris_explorer_amd/libephemeris_amd_exp_early.so python scripts/time_path.py 4096 20000 0 >> $O/time.log 2>&1
done
EPH_AMD_LIBRARY=$PWD/ephemeris_explorer_amd/libephemeris_amd_exp_early.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1
head -2 $O/pytest.log
cat $O/time.log
