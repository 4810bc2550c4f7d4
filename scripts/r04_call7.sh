#!/bin/bash
# round 4: after the split into per-order translation units (run-time pair variant) -- the whole GPU suite + the step timings
set -u
cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/r04h; mkdir -p $OUT
rm -f gpurun_out/pair_variant_times.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -n 15 $OUT/pytest.txt
cat gpurun_out/pair_variant_times.txt 2>/dev/null
python scripts/time_sizes.py 512 1024 2048 4096 8192 > $OUT/time_sizes.txt 2>&1; cat $OUT/time_sizes.txt
for k in 0 4 5 6; do EPH_PAIR_VARIANT=$k python scripts/time_path.py 4096 2000 0 2>&1 | tail -1; done | tee $OUT/variants_4096.txt
