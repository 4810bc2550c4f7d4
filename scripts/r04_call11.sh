#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/r04l; mkdir -p $OUT
T="timeout -k 5 -s KILL"
$T 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -n 8 $OUT/pytest.txt
$T 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
$T 200 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench_driver_form.json')); print(d['value'], d['ms_per_step'], d['blocks'], d['long_region']['ms_per_step'], d['other_configs']['configs3_craft_sweep']['value'])"
