#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/r04m; mkdir -p $OUT
T="timeout -k 5 -s KILL"
$T 400 python -m pytest tests/test_gpu_craft.py -x -q -m gpu > $OUT/pytest.txt 2>&1; tail -n 4 $OUT/pytest.txt
$T 150 python bench.py --workload craft --steps 5 --no-cpu-baseline > $OUT/bench_craft.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/bench_craft.json')); print(d['value'], d['ms_per_step'], d['roofline']['launch_us'], d['wall_over_kernel'])"
