#!/bin/bash
# round 4, first GPU call: (1) where the sweep's summary() spends its host time, dealt and not; (2) why a 256-gang of k_lm_small
# steps slower than one system; (3) the tile sweep at HEAD (all layouts still in the product library)
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04a; mkdir -p $OUT
export TMPDIR=/tmp
( for mode in 1 2; do for how in fresh reuse; do for tr in 1 2; do
    echo "== EPH_CRAFT_SORT=$mode $how EPH_TRACE_SUMMARY=$tr"
    EPH_CRAFT_SORT=$mode EPH_TRACE_SUMMARY=$tr timeout 300 python scripts/time_sweep_parts2.py 262144 $how 2>&1
  done; done; done ) > $OUT/sweep_parts.txt 2>&1
( echo "== forward only"; EPH_DEBUG_SMALL=4 EPH_DEBUG_PLACEMENT=1 timeout 300 python scripts/time_gang2.py 1 2 16 64 128 256 512 2>&1
  echo "== forward / backward alternating (round 3's gang)"; GANG_MIXED=1 timeout 200 python scripts/time_gang2.py 1 2 16 256 2>&1
  echo "== one workgroup per CU (EPH_SMALL_LDS_PAD=65536)"; EPH_SMALL_LDS_PAD=65536 EPH_DEBUG_SMALL=4 EPH_DEBUG_PLACEMENT=1 timeout 300 python scripts/time_gang2.py 1 16 128 256 512 2>&1
  echo "== per-phase ticks (smallacct build)"; EPH_AMD_LIBRARY=$PWD/ephemeris_explorer_amd/libephemeris_amd_exp_smallacct.so EPH_DEBUG_SMALL=4 timeout 300 python scripts/time_gang2.py 1 16 256 2>&1
  echo "== per-phase ticks, one workgroup per CU"; EPH_SMALL_LDS_PAD=65536 EPH_AMD_LIBRARY=$PWD/ephemeris_explorer_amd/libephemeris_amd_exp_smallacct.so EPH_DEBUG_SMALL=4 timeout 300 python scripts/time_gang2.py 256 2>&1
) > $OUT/gang.txt 2>&1
timeout 900 python scripts/tile_sweep3.py $PWD/ephemeris_explorer_amd/libephemeris_amd.so > $OUT/tile_sweep.jsonl 2> $OUT/tile_sweep.err
tail -n 60 $OUT/sweep_parts.txt; cat $OUT/gang.txt | grep -v "^  xcc" | tail -n 40; cat $OUT/tile_sweep.jsonl
