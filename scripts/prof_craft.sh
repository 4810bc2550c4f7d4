#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/prof_craft.sh <tag> [n_craft] [days]
# SQ / cache counters of the massless sweep kernel only (--kernel-include-regex) -> gpurun_out/prof_<tag>/
set -u
TAG=$1; N=${2:-262144}; DAYS=${3:-0.25}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/scripts/bench_craft.py $N $DAYS"
$CMD > $OUT/bench.json 2> $OUT/bench.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- $CMD > $OUT/stats.log 2>&1
timeout 200 rocprofv3 --kernel-trace --kernel-include-regex k_craft --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --kernel-include-regex k_craft --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum -d $OUT -o pmc2 -- $CMD > $OUT/pmc2.log 2>&1
rm -f $OUT/*_agent_info.csv
cat $OUT/bench.json | cut -c1-300
python - <<PY
import csv, glob, collections, json
out = {}
for f in sorted(glob.glob("$OUT/pmc*_counter_collection.csv")):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "k_craft_propagate" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
    out.update(acc)
for r in csv.DictReader(open("$OUT/stats_kernel_stats.csv")):
    if "k_craft" in r["Name"]: out["kernel"] = r["Name"]; out["calls"] = r["Calls"]; out["avg_ns"] = r["AverageNs"]
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out))
PY
