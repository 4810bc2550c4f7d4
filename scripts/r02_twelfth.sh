#!/bin/bash
# seeded reciprocal + leaner pair-wave loop: exactness sweep, parity, timing against the previous sequences
mkdir -p gpurun_out/r02l
O=gpurun_out/r02l
python - > $O/sweep.log 2>&1 <<'PY'
import time
import ephemeris_explorer_amd as ea
tot = 0
t = time.time()
for seed in range(1, 17):
    bad, ex = ea.debug_inv_r3_sweep(seed * 0x1234567, 1 << 34)
    tot += 1 << 34
    print(f"seed {seed}: {bad} mismatches of {1 << 34} (example bits {ex:#x})", flush=True)
print(f"{tot} operands in {time.time() - t:.1f} s")
PY
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_horizon.py tests/test_gpu_fast.py -x -q > $O/pytest.log 2>&1
for i in 1 2; do
python scripts/time_path.py 4096 20000 0 >> $O/time.log 2>&1
EPH_AMD_LIBRARY=$PWD/ephemeris_explorer_amd/libephemeris_amd_exp_noseed.so python scripts/time_path.py 4096 20000 0 >> $O/time.log 2>&1
done
python scripts/time_path.py 4096 20000 4 >> $O/time.log 2>&1
EPH_AMD_LIBRARY=$PWD/ephemeris_explorer_amd/libephemeris_amd_exp_noseed.so python scripts/time_path.py 4096 20000 4 >> $O/time.log 2>&1
python scripts/time_path.py 2048 20000 0 >> $O/time.log 2>&1
python scripts/time_path.py 8000 5000 0 >> $O/time.log 2>&1
cat $O/sweep.log | tail -4; tail -3 $O/pytest.log; cat $O/time.log
