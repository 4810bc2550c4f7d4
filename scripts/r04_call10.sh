#!/bin/bash
# every command under a hard time limit (a device-side deadlock must cost seconds, not the GPU budget)
set -u
cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/r04k; mkdir -p $OUT
L=$PWD/ephemeris_explorer_amd
T="timeout -k 5 -s KILL"
$T 60 python -c "
import numpy as np, ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.workloads import plummer
pos, vel, mu = plummer(4096)
a = ea.accel_eval(pos, mu); print('accel 4096 ok', float(np.abs(a).max()))
g = ea.NBodyIntegration(pos, vel, mu, 0.0, 1/1024); g.advance(12+20); g.sync(); print('steps ok')
" 2>&1 | tail -3
$T 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "accel or kernel_choice or every_tile or qt12 or other_methods" > $OUT/pytest.txt 2>&1; tail -n 4 $OUT/pytest.txt
for rep in 1 2; do for lib in libephemeris_amd.so libephemeris_amd_exp_barrier.so; do $T 60 python scripts/ab_step.py $L/$lib; done; done 2>&1 | tee $OUT/ab.txt
