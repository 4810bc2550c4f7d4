#!/bin/bash
# round 5: the step kernel's bounded experiments -- alternate the product library and the experimental builds named on the command
# line on ONE box (scripts/ab_step.py: HIP events after a one-second pre-warm, five blocks of 500 steps, N = 4096 and 2048), then
# the bit-parity tests on each experimental build.   usage: scripts/r05_step.sh NAME[:VAR=VALUE]...   (the variable is set for that
# build's runs only, e.g. duo:EPH_WG_BODIES=9)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
LOG=gpurun_out/step_ab.log; : > $LOG
P=$PWD/ephemeris_explorer_amd
for rep in 1 2; do
  timeout -k 5 120 python scripts/ab_step.py $P/libephemeris_amd.so >> $LOG 2>&1
  for spec in "$@"; do
    n=${spec%%:*}; e=${spec#*:}; [ "$e" = "$spec" ] && e="EPH_NOTHING=1"
    echo "-- $spec" >> $LOG
    env "$e" timeout -k 5 120 python scripts/ab_step.py $P/libephemeris_amd_exp_$n.so >> $LOG 2>&1
  done
done
for spec in "$@"; do
  n=${spec%%:*}; e=${spec#*:}; [ "$e" = "$spec" ] && e="EPH_NOTHING=1"
  echo "== parity on $spec" >> $LOG
  env "$e" EPH_AMD_LIBRARY=$P/libephemeris_amd_exp_$n.so timeout -k 5 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_horizon.py -m gpu -q -x \
      -k "accel or kernel_choice or every_tile or qt12 or other_methods or plummer or config5 or full_size" 2>&1 | grep -E "passed|failed|error" | tail -3 >> $LOG
done
