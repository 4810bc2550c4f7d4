#!/bin/bash
# On the GPU box: what the driver runs at the end of a round -- the -m gpu suite, smoke(), the default craft line
# (usage: scripts/final_check.sh TAG; output under gpurun_out/TAG)
set -u
cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
T="timeout -k 5 -s KILL"
$T 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1
grep -E "passed|failed|error" $OUT/pytest.txt | tail -3
$T 120 python -c 'import __graft_entry__ as g; g.smoke(); print("__SMOKE_OK__")' 2>&1 | tail -2 | tee $OUT/smoke.txt
$T 150 python bench.py --workload craft --steps 5 > $OUT/bench_craft.json 2> $OUT/bench_craft.err
tail -c 600 $OUT/bench_craft.json
