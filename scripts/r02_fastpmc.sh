#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT -o fast_pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --path fast --no-cpu-baseline --horizon 0 > $OUT/fast_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT -o fast_pmc_write -- python $GRAFT_REPO_ROOT/bench.py --path fast --no-cpu-baseline --horizon 0 > $OUT/fast_pmc_write.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $OUT -o fast_pmc_sq -- python $GRAFT_REPO_ROOT/bench.py --path fast --no-cpu-baseline --horizon 0 > $OUT/fast_pmc_sq.log 2>&1
rm -f $OUT/*_agent_info.csv
ls $OUT | grep fast_pmc
