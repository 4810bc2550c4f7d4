#!/bin/bash
# On the GPU box: the default bench + its rocprofv3 summaries for profiles/, then the round's other bench lines and timing
# scripts (usage: scripts/round_profile.sh r03; python scripts/summarize_profile.py r03 afterwards, in the container)
set -u
TAG=$1
T="timeout -k 5 -s KILL"     # every command under a hard limit: a device-side hang must cost seconds
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
$T 150 python bench.py > $OUT/bench.json 2> $OUT/bench.err
$T 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --horizon 0 > $OUT/bench_20a.json 2>/dev/null     # the driver's form, twice:
$T 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --horizon 0 > $OUT/bench_20b.json 2>/dev/null     # repeatability of the median block
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --horizon 0 --no-other-configs"   # the headline kernels only under the profiler
$T 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- $CMD > $OUT/stats.log 2>&1
$T 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $OUT -o pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
$T 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT -o pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
$T 200 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT -o pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
if [ "${LITE:-0}" = "1" ]; then          # (usage: LITE=1 scripts/round_profile.sh r03 -- the default line and its counters only)
  rm -f $OUT/*_agent_info.csv $OUT/*kernel_trace.csv.bak $OUT/*_kernel_trace.csv
  cat $OUT/bench.json
  exit 0
fi
# the other bench lines of the round + their kernel stats
cd $GRAFT_REPO_ROOT
$T 150 python bench.py --path fast > $OUT/bench_fast.json 2> $OUT/bench_fast.err
$T 150 python bench.py --path fast-rsq > $OUT/bench_fast_rsq.json 2> $OUT/bench_fast_rsq.err
$T 150 python bench.py --path f32-pairs > $OUT/bench_f32pairs_4096.json 2> $OUT/bench_f32pairs_4096.err
$T 150 python bench.py --bodies 65536 --path f32-pairs --steps 20 --warmup 3 > $OUT/bench_f32pairs.json 2> $OUT/bench_f32pairs.err
$T 150 python bench.py --workload craft --steps 3 > $OUT/bench_craft.json 2> $OUT/bench_craft.err
$T 240 python bench.py --workload craft --craft 1048576 --steps 1 --warmup 1 > $OUT/bench_craft_1m.json 2> $OUT/bench_craft_1m.err     # SURVEY 8(d)'s full width
$T 150 python bench.py --workload craft --population mixed --craft 524288 --craft-days 2 --steps 2 > $OUT/bench_craft_mixed.json 2> $OUT/bench_craft_mixed.err
$T 150 python bench.py --workload nbody-sharded --steps 20 --warmup 3 > $OUT/bench_sharded.json 2> $OUT/bench_sharded.err
# configs[4] as stated (binary32 pair arithmetic on a target partition): two ranks sharing this device, direct-write transport
EPH_BENCH_BACKEND=gloo $T 200 python bench.py --gpus 2 --workload nbody-sharded --path f32-pairs --transport peer --steps 20 --warmup 3 > $OUT/bench_f32pairs_sharded_gpus2_shared_device.json 2> $OUT/bench_f32pairs_sharded.err
# N > 1 flows on this one-GPU box: bench.py launches its own ranks; they share the device (gloo for the timing reductions),
# the sharded_4096 figure uses the direct-write transport between the two processes
EPH_BENCH_BACKEND=gloo $T 150 python bench.py --gpus 2 --steps 50 --warmup 5 --prewarm 0.5 > $OUT/bench_gpus2_shared_device.json 2> $OUT/bench_gpus2.err
$T 150 python scripts/time_small.py > $OUT/time_small.txt 2>&1
$T 150 python scripts/time_sizes.py 512 1024 2048 4096 8192 16384 > $OUT/time_sizes.txt 2>&1
cd /tmp
$T 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o fast_stats -- python $GRAFT_REPO_ROOT/bench.py --path fast --no-cpu-baseline --horizon 0 > $OUT/fast_stats.log 2>&1
$T 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o craft_stats -- python $GRAFT_REPO_ROOT/bench.py --workload craft --steps 3 --no-cpu-baseline > $OUT/craft_stats.log 2>&1
$T 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o f32_stats -- python $GRAFT_REPO_ROOT/bench.py --bodies 65536 --path f32-pairs --steps 20 --warmup 3 --no-cpu-baseline > $OUT/f32_stats.log 2>&1
rm -f $OUT/*_agent_info.csv $OUT/*kernel_trace.csv.bak $OUT/*_kernel_trace.csv
ls -la $OUT | head -50
cat $OUT/bench.json
