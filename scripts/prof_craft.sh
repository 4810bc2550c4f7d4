#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/prof_craft.sh <tag> [n_craft] [days]
# The massless sweep kernel only (--kernel-include-regex k_craft): kernel stats, SQ counters, cache requests, and the HBM-side bytes
# (FETCH_SIZE / WRITE_SIZE in passes of their own, as MI355X_MICROARCH.md prescribes) -> gpurun_out/prof_<tag>/summary.json
set -u
TAG=$1; N=${2:-262144}; DAYS=${3:-0.25}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/scripts/bench_craft.py $N $DAYS"
T="timeout -k 5 -s KILL 240"
$T $CMD > $OUT/bench.json 2> $OUT/bench.err
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- $CMD > $OUT/stats.log 2>&1
$T rocprofv3 --kernel-trace --kernel-include-regex k_craft --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1
$T rocprofv3 --kernel-trace --kernel-include-regex k_craft --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum -d $OUT -o pmc2 -- $CMD > $OUT/pmc2.log 2>&1
$T rocprofv3 --kernel-trace --kernel-include-regex k_craft --output-format csv --pmc FETCH_SIZE -d $OUT -o pmc3 -- $CMD > $OUT/pmc3.log 2>&1
$T rocprofv3 --kernel-trace --kernel-include-regex k_craft --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT -o pmc4 -- $CMD > $OUT/pmc4.log 2>&1
$T rocprofv3 --kernel-trace --kernel-include-regex k_craft --output-format csv --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d $OUT -o pmc5 -- $CMD > $OUT/pmc5.log 2>&1
rm -f $OUT/*_agent_info.csv
cut -c1-400 $OUT/bench.json
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
    if "k_craft_propagate" in r["Name"]: out["kernel"] = r["Name"]; out["calls"] = r["Calls"]; out["avg_ns"] = r["AverageNs"]
try:
    out["bench"] = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
except Exception as e:
    out["bench_error"] = str(e)
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "bench"}))
PY
