#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/prof.sh <tag> <run_case args...>
# writes gpurun_out/prof_<tag>/{stats,pmc1,pmc2}*  (kernel-trace+stats in one run, counters in their own runs)
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python $GRAFT_REPO_ROOT/scripts/run_case.py "$@" > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $OUT -o pmc1 -- python $GRAFT_REPO_ROOT/scripts/run_case.py "$@" > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU -d $OUT -o pmc2 -- python $GRAFT_REPO_ROOT/scripts/run_case.py "$@" > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT -o pmc3 -- python $GRAFT_REPO_ROOT/scripts/run_case.py "$@" > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT -o pmc4 -- python $GRAFT_REPO_ROOT/scripts/run_case.py "$@" > $OUT/pmc4.log 2>&1
ls $OUT | head -40
tail -2 $OUT/*.log
