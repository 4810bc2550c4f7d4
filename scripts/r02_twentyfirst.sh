#!/bin/bash
# under layout 5: pinned chain tile (EPH_CHAIN_ASM) and the SGPR-base source loads (EPH_PAIR_LOOP=1) against the default
mkdir -p gpurun_out/r02u
O=gpurun_out/r02u
for i in 1 2; do
python scripts/time_path.py 4096 20000 0 >> $O/time.log 2>&1
for v in chainasm loop1; do
EPH_AMD_LIBRARY=$PWD/ephemeris_explorer_amd/libephemeris_amd_exp_$v.so python scripts/time_path.py 4096 20000 0 >> $O/time.log 2>&1
done
done
EPH_AMD_LIBRARY=$PWD/ephemeris_explorer_amd/libephemeris_amd_exp_chainasm.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest_chainasm.log 2>&1
head -2 $O/pytest_chainasm.log
cat $O/time.log
