#!/bin/bash
# the mixed population (dealt / undealt static / undealt queue) on the product and the named experimental builds, one box
set -u
cd $GRAFT_REPO_ROOT
L=$PWD/ephemeris_explorer_amd
for lib in libephemeris_amd.so "$@"; do
  for e in "EPH_CRAFT_SORT=1" "EPH_CRAFT_SORT=0 EPH_CRAFT_QUEUE=1"; do
    env $e EPH_AMD_LIBRARY=$L/$lib timeout 300 python bench.py --workload craft --population mixed --craft 524288 --craft-days 2 --steps 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"$lib $e\", d[\"value\"], d[\"ms_per_step\"], d[\"kernel_ms_rank0\"])"
  done
done
