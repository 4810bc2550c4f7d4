#!/bin/bash
# round 5: k_lm_small after a change -- bit parity of the small-system tests, the per-phase tick accounting (tuning build
# -DEPH_EXPERIMENTS=1 -DEPH_SMALL_ACCOUNT=1; EPH_DEBUG_SMALL=4, 5 = without the pair phase's LDS stores), the timings
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_gang.py tests/test_gpu_variants.py -m gpu -q -x -k "qt12_state or other_methods or degenerate or single_steps or gang or many or 1e5_steps_bitwise or variant" 2>&1 | tail -3 > gpurun_out/small_parity.log
L=$PWD/ephemeris_explorer_amd/libephemeris_amd_exp_smallacct.so
rm -f gpurun_out/small_clock.log
if [ -f $L ]; then
  for f in ${SMALL_FLAGS:-4}; do
    echo "== EPH_DEBUG_SMALL=$f" >> gpurun_out/small_clock.log
    EPH_AMD_LIBRARY=$L EPH_DEBUG_SMALL=$f python scripts/clock_small.py >> gpurun_out/small_clock.log 2>&1
  done
fi
python scripts/time_small.py > gpurun_out/small_time.log 2>&1
