#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/prof_craft.sh <tag> [n_craft] [days]
# SQ / cache counters of the massless sweep kernel -> gpurun_out/prof_<tag>/
set -u
TAG=$1; N=${2:-65536}; DAYS=${3:-0.25}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --workload craft --craft $N --craft-days $DAYS --steps 2 --warmup 1"
$CMD > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM -d $OUT -o pmc2 -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT -o pmc3 -- $CMD > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE WRITE_SIZE -d $OUT -o pmc4 -- $CMD > $OUT/pmc4.log 2>&1
rm -f $OUT/*_agent_info.csv
ls $OUT
cat $OUT/bench.json
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/pmc*_counter_collection.csv")):
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "k_craft_propagate" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    print(f.split("/")[-1], {k: (v, cnt[k]) for k, v in acc.items()})
PY
