#!/bin/bash
# On the GPU box: the configs[2] sweep over the force kernel's tiling (bodies per wave / workgroup form) at N = 4096.
# The source tile is one wave64 (64 bodies) in every variant; what varies is how many target bodies share a tile's
# ordered-sum phase. usage: scripts/tile_sweep.sh <tag>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
: > $OUT/tile_sweep.jsonl
for v in "wave 1" "wave 2" "wave 4" "wave 8" "wg 0"; do
  set -- $v
  EPH_FORCE=$1 EPH_BPW=$2 python bench.py --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | tail -1 | \
    python -c "import sys, json; d = json.loads(sys.stdin.read()); print(json.dumps({'force': '$1', 'bodies_per_wave': $2, 'us_per_step': d['roofline']['launch_us'], 'body_steps_per_s': d['value'], 'fp64_frac': d['fp64']['frac']}))" >> $OUT/tile_sweep.jsonl
done
cat $OUT/tile_sweep.jsonl
