#!/bin/bash
# On the GPU box: the default bench + its rocprofv3 summaries for profiles/ (usage: scripts/round_profile.sh r01)
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --horizon 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $OUT -o pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT -o pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT -o pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
# the other bench lines of the round (fast path, massless sweep, one large sharded system) + their kernel stats
cd $GRAFT_REPO_ROOT
python bench.py --path fast > $OUT/bench_fast.json 2> $OUT/bench_fast.err
python bench.py --path fast-rsq > $OUT/bench_fast_rsq.json 2> $OUT/bench_fast_rsq.err
python bench.py --workload craft --steps 3 > $OUT/bench_craft.json 2> $OUT/bench_craft.err
python bench.py --workload nbody-sharded --steps 20 --warmup 3 > $OUT/bench_sharded.json 2> $OUT/bench_sharded.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o fast_stats -- python $GRAFT_REPO_ROOT/bench.py --path fast --no-cpu-baseline --horizon 0 > $OUT/fast_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o craft_stats -- python $GRAFT_REPO_ROOT/bench.py --workload craft --steps 3 --no-cpu-baseline > $OUT/craft_stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $OUT -o fast_pmc_sq -- python $GRAFT_REPO_ROOT/bench.py --path fast --no-cpu-baseline --horizon 0 > $OUT/fast_pmc_sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT -o fast_pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --path fast --no-cpu-baseline --horizon 0 > $OUT/fast_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT -o fast_pmc_write -- python $GRAFT_REPO_ROOT/bench.py --path fast --no-cpu-baseline --horizon 0 > $OUT/fast_pmc_write.log 2>&1
rm -f $OUT/*_agent_info.csv $OUT/*kernel_trace.csv.bak
ls -la $OUT | head -30
cat $OUT/bench.json
