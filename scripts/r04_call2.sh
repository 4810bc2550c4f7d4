#!/bin/bash
# round 4, second GPU call: the sweep's read-back without the copy engine + cached knot slabs; per-workgroup times of a gang
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04b; mkdir -p $OUT
export TMPDIR=/tmp
( for how in fresh reuse; do for tr in 1 2; do
    echo "== EPH_CRAFT_SORT=2 $how EPH_TRACE_SUMMARY=$tr"
    EPH_TRACE_SUMMARY=$tr timeout 300 python scripts/time_sweep_parts2.py 262144 $how 2>&1
  done; done
  echo "== pool off (EPH_POOL_MAX_MB=0), fresh"; EPH_POOL_MAX_MB=0 EPH_TRACE_SUMMARY=1 timeout 300 python scripts/time_sweep_parts2.py 262144 fresh 2>&1
  echo "== not dealt (EPH_CRAFT_SORT=1), fresh"; EPH_CRAFT_SORT=1 EPH_TRACE_SUMMARY=1 timeout 300 python scripts/time_sweep_parts2.py 262144 fresh 2>&1
) > $OUT/sweep_parts.txt 2>&1
( echo "== forward only"; EPH_DEBUG_SMALL=4 EPH_DEBUG_PLACEMENT=1 timeout 300 python scripts/time_gang2.py 1 16 128 256 2>&1
  echo "== per workgroup, 256"; EPH_DEBUG_SMALL=4 EPH_DEBUG_PLACEMENT=2 timeout 300 python scripts/time_gang2.py 256 2>&1
  echo "== per-phase ticks (smallacct build)"; EPH_AMD_LIBRARY=$PWD/ephemeris_explorer_amd/libephemeris_amd_exp_smallacct.so EPH_DEBUG_SMALL=4 timeout 300 python scripts/time_gang2.py 1 16 256 2>&1
) > $OUT/gang.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_craft.py tests/test_gpu_cli.py tests/test_gpu_bench.py -x -q -m gpu > $OUT/pytest_craft.txt 2>&1
python bench.py --workload craft --steps 3 > $OUT/bench_craft.json 2> $OUT/bench_craft.err
tail -n 45 $OUT/sweep_parts.txt; grep -v "^  wg " $OUT/gang.txt | tail -n 40; tail -n 5 $OUT/pytest_craft.txt; cat $OUT/bench_craft.json | cut -c1-600
