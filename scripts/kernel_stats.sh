#!/bin/bash
# On the GPU box: rocprofv3 kernel-trace summaries of the default line and the craft line only (usage: scripts/kernel_stats.sh TAG;
# scripts/round_profile.sh takes the counters as well)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
T="timeout -k 5 -s KILL"
cd /tmp && export TMPDIR=/tmp
$T 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --horizon 0 > $OUT/stats.log 2>&1
$T 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o craft_stats -- python $GRAFT_REPO_ROOT/bench.py --workload craft --steps 3 --no-cpu-baseline > $OUT/craft_stats.log 2>&1
rm -f $OUT/*_agent_info.csv $OUT/*_kernel_trace.csv
ls -la $OUT; head -5 $OUT/*stats_kernel_stats.csv | cut -c1-200
