#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/r04j; mkdir -p $OUT
timeout 2700 python -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -n 12 $OUT/pytest.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench.json'))
print({k: d[k] for k in ('value','ms_per_step','blocks','pair_variant')}, d['long_region'], d['roofline']['fp64']['frac'], d['roofline']['launch_us'])
print(d['other_variants']); print(d['other_configs'])"
