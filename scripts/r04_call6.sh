#!/bin/bash
# round 4: after the sweep / gang / peer changes -- GPU tests of the touched areas + the timing scripts
set -u
cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/r04g; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_craft.py tests/test_gpu_cli.py tests/test_gpu_bench.py tests/test_gpu_gang.py tests/test_gpu_shard.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -n 8 $OUT/pytest.txt
python scripts/time_sweep_parts2.py 262144 fresh > $OUT/sweep_parts.txt 2>&1; grep -v "^summary" $OUT/sweep_parts.txt
python scripts/time_gang2.py 1 2 16 128 256 512 1024 > $OUT/gang.txt 2>&1; cat $OUT/gang.txt
python scripts/time_small.py > $OUT/time_small.txt 2>&1; cat $OUT/time_small.txt
python bench.py --workload craft --steps 3 > $OUT/bench_craft.json 2> $OUT/bench_craft.err; cut -c1-400 $OUT/bench_craft.json
